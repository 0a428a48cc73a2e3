"""bench.py's sliced-network leg alone (64-node 3-regular network, bond 12, bf16, all 144 slices, every slice partial
checked against the same slice in f32 by bench.partials_check), with whatever lowering the environment selects
(TNH_GATHER_GEMM=0/1, TNH_GATHER_MIN_PIECE).
  python tools/rr64_check.py [--D 12]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensornetwork_amd as ta
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--D", type=int, default=12)
a = ap.parse_args()
be = ta.get_hip_backend()
rec = bench.sliced_network_bench(ta, be, None, 0, 1, a.D, 64, True)
rec["gather_launches"] = be.gather_launches
rec["permute_launches"] = be.permute_launches
print(json.dumps({k: rec.get(k) for k in ("seconds", "tflops", "mode", "flops_total", "flops_if_every_slice_ran_alone", "n_slices", "result",
                                          "verified", "gather_launches", "permute_launches")}))
