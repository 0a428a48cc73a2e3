"""Complex-dtype rows on the CPU: pins the oracle's svd / qr / rq / eigh / inv / expm
against outputs of the reference itself on complex64 / complex128 inputs (tests/golden/golden_complex.npz)
through the same checks the GPU suite applies to HipBackend."""
import numpy as np

from oracle import numpy_oracle as orc
import cases as C


def test_oracle_complex_svd_matches_reference(golden_complex):
  be = orc.OracleBackend()
  for case in golden_complex.cases["svd"]:
    C.check_svd_case(be, golden_complex, case)


def test_oracle_complex_qr_rq_match_reference(golden_complex):
  be = orc.OracleBackend()
  for case in golden_complex.cases["qr"]:
    C.check_qr_case(be, golden_complex, case)


def test_oracle_complex_eigh_inv_expm_match_reference(golden_complex):
  be = orc.OracleBackend()
  for case in golden_complex.cases["linalg"]:
    C.check_linalg_case(be, golden_complex, case)
