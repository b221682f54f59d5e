"""Print VGPR/SGPR/spill/scratch/occupancy of every kernel in a HIP source (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/kernel_resources.py vs_seg_amd/csrc/wgrad.hip [filter]
    python tools/kernel_resources.py vs_seg_amd/csrc/igemm_inst.hip "" -DIG_T=bf16_t -DIG_TNAME=bf16 -DIG_NT=2
"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
extra = [a for a in sys.argv[3:] if a.startswith("-")]
cmd = ["/opt/rocm/bin/hipcc", *extra, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Iinclude", "-I../../include", "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in err.splitlines():
    m = re.search(r"remark:\s+(.*?):\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
print(f"{'kernel':60s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'vspill':>6s} {'sspill':>6s} {'scratch':>7s} {'occ':>4s}")
for r in rows:
    if flt in r["name"]:
        print(f"{r['name'][:60]:60s} {r.get('VGPRs', ''):>5s} {r.get('AGPRs', ''):>5s} {r.get('TotalSGPRs', ''):>5s} {r.get('VGPRs Spill', ''):>6s} {r.get('SGPRs Spill', ''):>6s} {r.get('ScratchSize [bytes/lane]', ''):>7s} {r.get('Occupancy [waves/SIMD]', ''):>4s}")
