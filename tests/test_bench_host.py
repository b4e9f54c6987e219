"""Host-side logic of bench.py that needs no GPU: the traffic table's stamps and the self-launcher."""
import json
import os
import subprocess
import sys

import bench
from conftest import ROOT


def _table(tmp_path, **over):
    entry = {"bytes": 123456789, "source": "r9_test", "kernel_src_sha": bench.kernel_source_sha("csr_stream_local"),
             "format_bytes": 1000, "blocks": 77}
    entry.update(over)
    path = tmp_path / "traffic.json"
    path.write_text(json.dumps({"csr_stream_local|some workload": entry}))
    return str(path)


def test_traffic_is_handed_out_only_for_the_profiled_kernel_and_plan(tmp_path):
    ok = _table(tmp_path)
    got, src = bench.measured_traffic("csr_stream_local", "some workload", 1000, 77, ok)
    assert got == 123456789 and "r9_test" in src and "not measured in this run" in src
    # another workload / kernel: nothing recorded
    assert bench.measured_traffic("csr_stream_local", "other workload", 1000, 77, ok)[0] is None
    assert bench.measured_traffic("csr_stream", "some workload", 1000, 77, ok)[0] is None
    # a re-planned kernel (other format bytes or workgroup count) gets null
    assert bench.measured_traffic("csr_stream_local", "some workload", 1001, 77, ok)[0] is None
    assert bench.measured_traffic("csr_stream_local", "some workload", 1000, 78, ok)[0] is None
    # a changed kernel source gets null, with the reason
    stale = _table(tmp_path, kernel_src_sha="0123456789abcdef")
    got, why = bench.measured_traffic("csr_stream_local", "some workload", 1000, 77, stale)
    assert got is None and "kernel source changed" in why
    # round-1 style bare numbers carry no stamp: null
    bare = tmp_path / "bare.json"
    bare.write_text(json.dumps({"csr_stream_local|some workload": 5}))
    assert bench.measured_traffic("csr_stream_local", "some workload", 1000, 77, str(bare))[0] is None


def test_committed_traffic_table_is_stamped():
    table = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    for key, e in table.items():
        assert isinstance(e, dict) and {"bytes", "source", "kernel_src_sha", "format_bytes", "blocks"} <= set(e), key


def test_self_launch_starts_child_ranks_and_relays_one_json_line(tmp_path):
    """`python bench.py --gpus 2` with no launcher in the environment must spawn the ranks itself
    (torch.distributed.run as a child) and exit with their code.  Without a GPU the ranks stop at
    "needs a HIP device": the launcher has to pass that failure on (non-zero, nothing on stdout)."""
    import sparsematrixvectormultiplication_amd as sp
    if sp.device_count() > 0:
        import pytest
        pytest.skip("a HIP device is present: covered by tests/test_c_driver.py on the GPU box")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1",
                           "--warmup", "0", "--grid", "8,8,8", "--exchange", "gloo-host", "--no-cpu-baseline"],
                          capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert proc.returncode != 0
    assert "starting 2 ranks" in proc.stderr and "--nproc-per-node=2" in proc.stderr
    assert "needs a HIP device" in proc.stderr
    assert proc.stdout.strip() == ""
