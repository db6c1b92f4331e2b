"""usage: python tools/kres.py <file.hip> [regex] [extra hipcc flags...]
Per kernel of one source file: VGPRs, AGPRs, spilled registers, occupancy (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re, subprocess, sys
f = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "."
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-unused-result",
       "-Rpass-analysis=kernel-resource-usage", *sys.argv[3:], "-c", f, "-o", "/tmp/_kres.o"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
keys = {"VGPRs": "v", "AGPRs": "a", "VGPRs Spill": "vspill", "SGPRs Spill": "sspill", "Occupancy [waves/SIMD]": "occ", "LDS Size [bytes/block]": "lds"}
for l in out.splitlines():
    m = re.search(r"Function Name: (\S+)", l)
    if m:
        cur = {"n": m.group(1)}
        rows.append(cur)
        continue
    for k, s in keys.items():
        m = re.search(r"    " + re.escape(k) + r": (\d+)", l)
        if m and cur is not None:
            cur[s] = m.group(1)
if not rows:
    print(out[-3000:])
for r in rows:
    n = subprocess.run(["c++filt", r["n"]], capture_output=True, text=True).stdout.strip()
    n = re.sub(r"\(anonymous namespace\)::", "", n).replace("void ", "")
    n = re.sub(r"\(.*", "", n)
    if re.search(pat, n):
        print(f"{n:64s} " + " ".join(f"{s}={r.get(s)}" for s in keys.values()))
