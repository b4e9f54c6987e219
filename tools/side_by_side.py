#!/usr/bin/env python3
"""CPU and GPU side by side: runs bench.py for the three single-GPU
configurations and prints one markdown row per matrix / format."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = []
for workload in ("cant", "cant_hll", "nlpkkt"):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--no-also"],
                       capture_output=True, text=True, cwd=ROOT)
    if p.returncode:
        print(p.stderr[-2000:])
        raise SystemExit(f"bench.py --workload {workload} failed")
    r = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    cb, rf = r["cpu_baseline"], r["roofline"]
    serial = cb.get("serial_csr_gflops_1core", cb.get("serial_hll_gflops_1core"))
    rows.append(f"| {r['config']['workload']} | {r['config']['nnz']} | {serial} | {cb['value']} ({cb['cores']} cores, "
                f"{cb['kind']}) | {r['value']} | {rf['kernel']} | {rf['kernel_ms_mean'] * 1e3:.1f} | {rf['achieved']} | "
                f"{rf['frac'] * 100:.1f} | {rf['frac_by_format_bytes'] * 100:.1f} | "
                f"{r['parity_vs_cpu_reference']['max_abs_diff_over_max_abs']:.1e} |")
print("| matrix / format | nnz | CPU serial GFLOP/s | CPU OpenMP GFLOP/s | GPU GFLOP/s | GPU kernel | kernel us | "
      "algorithmic GB/s | % of 8 TB/s | % by format bytes | max abs diff / max abs y (GPU vs CPU) |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
print("\n".join(rows))
