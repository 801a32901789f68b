"""Instruction mix of one kernel in a hipcc -S --cuda-device-only listing: vector / scalar / LDS / global counts, the whole
kernel and the part after the first s_barrier (the plane loop of the z-streaming kernels).
    hipcc --offload-arch=gfx950 -O3 ... -S --cuda-device-only -o /tmp/k.s csrc/conv_c8.hip; python tools/asm_mix.py /tmp/k.s <name substring> [top]"""
import collections
import sys

path, pat = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 0
lines = open(path).read().splitlines()
i = 0
while i < len(lines):
    if lines[i].startswith("_Z") and pat in lines[i].split(":")[0]:
        name = lines[i].split(":")[0]
        j = i + 1
        body = []
        while j < len(lines) and "s_endpgm" not in lines[j]:
            t = lines[j].strip()
            if t and not t.startswith((";", ".")) and not t.endswith(":"):
                body.append(t.split()[0])
            j += 1
        first = next((k for k, op in enumerate(body) if op == "s_barrier"), 0)
        for tag, ops in (("kernel", body), ("after the first barrier", body[first:])):
            c = collections.Counter(op.split("_")[0] for op in ops)
            m = sum(1 for op in ops if op.startswith("v_mfma"))
            print("%s | %s: %d vector (%d mfma) %d scalar %d ds %d global" % (name[:140], tag, c["v"], m, c["s"], c["ds"], c["global"] + c["buffer"]))
        if top:
            print("   ", collections.Counter(body[first:]).most_common(top))
        i = j
    i += 1
