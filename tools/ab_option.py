"""A/B of one engine option inside the real benchmark step: python tools/ab_option.py <option> <value> -> ms per step."""
import contextlib, io, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
name, v = sys.argv[1], int(sys.argv[2])
sys.argv = ['bench.py', '--headline-only', '--no-cpu-baseline', '--steps', '20', '--warmup', '3']
from reprover_amd import _lib
_lib.check(_lib.load().rp_set_option(name.encode(), v), "rp_set_option")
import bench
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads(buf.getvalue().strip().splitlines()[-1])
k = d["kernel_ms_per_step"]
print(name, v, round(d["ms_per_step"], 3), {x: round(k[x], 3) for x in ("gemm_qkv", "gemm_o", "gemm_wi", "gemm_wo", "attention")})
