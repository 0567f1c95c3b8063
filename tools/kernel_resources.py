"""Register / spill / LDS table of one HIP source: python tools/kernel_resources.py rp_encoder.hip [name-filter]"""
import os, re, subprocess, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = src if os.path.exists(src) else os.path.join(here, "reprover_amd", "csrc", src)
extra = os.environ.get("EXTRA", "").split()
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", *extra, "-c", path, "-o", "/dev/null",
                    "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, cwd="/tmp")
cur = None
rows = []
for line in r.stderr.splitlines():
    m = re.search(r"remark: (?:\S+ )?\s*(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs|LDS Size \[bytes/block\]|VGPRs Spill|SGPRs Spill): (.*?) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    elif cur is not None:
        cur[k.split(" ")[0] if "Spill" not in k else k] = v
names = subprocess.run(["c++filt"], input="\n".join(x["name"] for x in rows), capture_output=True, text=True).stdout.splitlines()
for x, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n).replace("rp::", "")
    if flt and flt not in n:
        continue
    print(f'{n[:150]:150s} v{x.get("VGPRs","?"):>4} a{x.get("AGPRs","?"):>4} s{x.get("SGPRs","?"):>4} scratch{x.get("ScratchSize","?"):>5} occ{x.get("Occupancy","?"):>2}')
if r.returncode != 0:
    print(r.stderr[-3000:])
