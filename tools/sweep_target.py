#!/usr/bin/env python3
"""Run a fixed list of stream-kernel configurations, N launches each, for rocprofv3 to
observe (tools/sweep_pmc.sh).  Writes the list to gpurun_out/sweep_configs.json."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsematrixvectormultiplication_amd as sp  # noqa: E402
from sparsematrixvectormultiplication_amd import synth  # noqa: E402
from sparsematrixvectormultiplication_amd.device import set_tuning  # noqa: E402

LAUNCHES = 6
CONFIGS = []
CONFIGS.append(("probe stream", 4096, dict(stream_kind=10, stream_xcd=0)))
CONFIGS.append(("probe stream+gather real x (28 MB)", 4096, dict(stream_kind=11, stream_xcd=0)))
for mask, nm in ((1023, "8 KiB"), (4095, "32 KiB"), (32767, "256 KiB"), (262143, "2 MiB"), (1048575, "8 MiB"), (2097151, "16 MiB")):
    CONFIGS.append((f"probe stream+gather table {nm}", 4096, dict(stream_kind=15, stream_xcd=0, probe_mask=mask)))


def main():
    sp.hip_init(0)
    M, row_ptr, col, val = synth.kkt_like()
    devs = {}
    for cap in (2048, 4096):
        set_tuning("stream_cap", cap)
        devs[cap] = sp.CsrDevice(M, M, row_ptr, col, val)
        devs[cap].set_x(np.ones(M))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump({"launches": LAUNCHES, "configs": [c[0] for c in CONFIGS],
               "algo_bytes": devs[2048].info()["algo_bytes"]}, open("gpurun_out/sweep_configs.json", "w"))
    for name, cap, knobs in CONFIGS:
        for k, v in knobs.items():
            set_tuning(k, v)
        for _ in range(LAUNCHES):
            devs[cap].run(sp.CSR_STREAM)
        sp.hip_sync()


if __name__ == "__main__":
    main()
