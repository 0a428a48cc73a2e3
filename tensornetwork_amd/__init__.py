"""tensornetwork_amd -- MI355X (gfx950) native backend for google/TensorNetwork.

``import tensornetwork_amd`` registers the backend name ``"hip"`` with
TensorNetwork's backend factory when that library is importable, so that
``tn.Node(x, backend="hip")``, ``tn.ncon(..., backend="hip")``,
``tn.contractors.greedy`` and ``tn.split_node`` run on the GPU unchanged.  The
same workloads also run stand-alone through ``tensornetwork_amd.ncon`` /
``.network`` / ``.contractors`` on a machine that only has this repository.

Compute lives in ``libtnhip.so`` (hand-written HIP kernels, C ABI in
``include/tnh.h``); there is no CPU fallback.
"""
from tensornetwork_amd._lib import HipRuntimeError
from tensornetwork_amd.device_tensor import DeviceTensor, bfloat16, round_to_bf16, configure_gc, gc_stats
from tensornetwork_amd.device_tensor import trim as trim_pool
from tensornetwork_amd.hip_backend import (HipBackend, get_hip_backend,
                                           register_with_tensornetwork)
from tensornetwork_amd.ncon import (ncon, einsum, DefaultBackend, set_default_backend,
                                    get_default_backend)
from tensornetwork_amd.network import (Node, Edge, NodeCollection, CopyNode, contract_copy_node, connect, contract, contract_between,
                                       contract_parallel, contract_trace_edges, outer_product,
                                       split_node, split_node_full_svd, split_node_qr, split_node_rq, copy,
                                       slice_edge,
                                       get_all_edges, get_subgraph_dangling, get_shared_edges,
                                       reachable, nodes_to_json, nodes_from_json,
                                       get_parallel_edges, get_all_nondangling, get_all_dangling,
                                       get_all_nodes, get_neighbors, check_connected, check_correct,
                                       disconnect, remove_node, redirect_edge, flatten_edges,
                                       flatten_edges_between, flatten_all_edges, split_edge,
                                       replicate_nodes, reduced_density, from_topology, switch_backend,
                                       outer_product_final_nodes)
from tensornetwork_amd import contractors, pathfinder
from tensornetwork_amd.tensor import Tensor, NconBuilder, finalize
from tensornetwork_amd import linalg, node_linalg
from tensornetwork_amd.decorators import jit
from tensornetwork_amd.mps import FiniteMPS, InfiniteMPS, FiniteDMRG
from tensornetwork_amd.mpo import (BaseMPO, FiniteMPO, InfiniteMPO, FiniteXXZ, FiniteTFI,
                                   FiniteFreeFermion2D)

__version__ = "0.1.0"

REGISTERED = register_with_tensornetwork()
