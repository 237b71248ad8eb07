"""Register / LDS / spill table of the kernels of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).

    python scripts/kernel_resources.py ieee_amd/csrc/conv.hip [name filter]
"""
import re
import subprocess
import sys


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "--cuda-device-only", "-c", src,
           "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in out.splitlines():
        m = re.search(r"remark: (?:\S+: )?\s*(.*?) \[-Rpass-analysis", line)
        if not m:
            continue
        body = m.group(1).strip()
        if body.startswith("Function Name:"):
            cur = {"name": body.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in body:
            k, v = body.split(":", 1)
            cur[k.strip()] = v.strip()
    demangle = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows),
                              capture_output=True, text=True).stdout.splitlines()
    print(f"{'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'spillV':>6} {'scratch':>7} {'occ':>3} {'LDS':>6}  kernel")
    for r, name in zip(rows, demangle):
        name = re.sub(r"\(.*", "", name).replace("void ieee::", "")
        if flt and flt not in name:
            continue
        print(f"{r.get('VGPRs', '?'):>5} {r.get('AGPRs', '?'):>5} {r.get('TotalSGPRs', '?'):>5} {r.get('VGPRs Spill', '?'):>6} "
              f"{r.get('ScratchSize [bytes/lane]', '?'):>7} {r.get('Occupancy [waves/SIMD]', '?'):>3} "
              f"{r.get('LDS Size [bytes/block]', '?'):>6}  {name}")


if __name__ == "__main__":
    main()
