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
for mode, name in ((0, "stream"), (1, "stream+gather"), (5, "stream+gather(L1 table)"), (3, "all"), (7, "all, gather(L1 table)")):
    CONFIGS.append((f"probe cap=4096 {name}", 4096, dict(stream_kind=10 + mode, stream_xcd=0)))
CONFIGS.append(("prod cap=4096", 4096, dict(stream_kind=0, stream_xcd=0)))
CONFIGS.append(("walk cap=4096", 4096, dict(stream_kind=1, stream_xcd=0)))


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
