"""gfx950 store-data hazard scan (round 5).

A 12- / 16-byte buffer store whose soffset is a scalar REGISTER still reads its data registers when the next vector instructions
issue; LLVM's hazard recognizer inserts the wait states only for an immediate soffset (GCNHazardRecognizer::createsVALUHazard),
and on gfx950 an instruction that overwrites those registers right behind the store puts its NEW values for lanes 12-15 of each
row of 16 (second dword) into memory.  That was the "nondeterministic" stride-2 fused conv-GRU cell with 8-row tiles
(csrc/gru_fused.hip, tools/gru2_debug.py); csrc/common.h buffer_store_b128_guarded is the fix.

scan_library() disassembles every code object of the built library (llvm-objdump) and reports each buffer_store_dwordx3 / x4 with
an SGPR soffset that is followed within WAIT instruction slots by a write to one of its data registers.  tests/test_tools.py runs
it on the production library (CPU).    python tools/store_hazard_scan.py [path/to/libdeep3d_planesweep.so]"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
WAIT = 2   # instruction slots that must separate the store from the overwrite (gfx940+: two wait states)
STORE = re.compile(r"^\s*buffer_store_dwordx[34]\s+v\[(\d+):(\d+)\],\s*(?:v\d+|v\[\d+:\d+\]|off),\s*s\[\d+:\d+\],\s*(\S+)")
NODEST = ("s_", "ds_write", "ds_store", "buffer_store", "global_store", "flat_store", "scratch_store", "v_cmp", "v_cmpx", "buffer_wbl2", "buffer_inv")


def dst_regs(text):
    t = text.split(None, 1)
    if len(t) < 2 or t[0].startswith(NODEST):
        return set()
    d = t[1].split(",")[0].strip()
    m = re.match(r"v\[(\d+):(\d+)\]", d)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", d)
    return {int(m.group(1))} if m else set()


def scan_listing(lines):
    """lines: disassembly text (objdump or `hipcc -S`).  Returns (wide stores with an SGPR soffset, [(kernel, store, overwrite, slots)])."""
    kern, hits, nstores = None, [], 0
    ins = []
    for l in lines:
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", l) or re.match(r"^(_Z\w+):", l)
        if m:
            kern = m.group(1)
            continue
        t = l.split("//")[0].split(";")[0].strip()
        if t and not t.startswith(".") and not t.endswith(":"):
            ins.append((kern, t))
    for i, (k, t) in enumerate(ins):
        m = STORE.match(t)
        if not m or not m.group(3).startswith("s"):
            continue
        nstores += 1
        data = set(range(int(m.group(1)), int(m.group(2)) + 1))
        seen, j = 0, i + 1
        while seen < WAIT and j < len(ins) and ins[j][0] == k:
            u = ins[j][1]
            j += 1
            if u.startswith("s_nop"):
                seen += 1 + int(u.split()[1], 0)
                continue
            if dst_regs(u) & data:
                hits.append((k, t, u, seen))
                break
            seen += 1
    return nstores, hits


def scan_library(path=None):
    sys.path.insert(0, ROOT)
    from deep3d_aerial_amd import _lib

    nstores, hits = 0, []
    for elf in _lib.code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as tmp:
            tmp.write(elf)
            tmp.flush()
            out = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", tmp.name], capture_output=True, text=True, check=True).stdout
        n, h = scan_listing(out.split("\n"))
        nstores += n
        hits += h
    return nstores, hits


if __name__ == "__main__":
    n, hits = scan_library(sys.argv[1] if len(sys.argv) > 1 else None)
    print("%d 12- / 16-byte buffer stores with a register soffset; %d overwritten within %d instruction slots" % (n, len(hits), WAIT))
    for k, a, b, s in hits:
        print("  %s\n      %s\n      +%d: %s" % (k[:120], a, s, b))
    sys.exit(1 if hits else 0)
