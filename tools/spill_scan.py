"""Static check of a gfx950 code object: scratch (spill) STORES that execute under a narrowed EXEC mask.

Root cause of the wrong iterates of the 128-VGPR sweep kernels (KMAX = 22 in round 2, KMAX = 20 in round 3): the register
allocator placed the spill store of a value that is live in ALL lanes (the gene's row offset j * KP) inside an exec-masked
region (`s_and_saveexec_b64` ... `s_or_b64 exec`, the `if (16 + i < K)` body of the second coordinate slot), so only the lanes
active there saved it; the reload after the loop ran under the full mask and the other lanes formed their store addresses
from garbage.  A scratch store under a narrowed mask is only right when every later reload runs under a subset of that mask;
the cheap, conservative check is: no scratch store inside a region whose mask was narrowed (linear scan of the listing:
saveexec / s_and exec open a region, `s_or_b64 exec, exec, ...` / `s_mov_b64 exec, -1` close it).

    python tools/spill_scan.py [libinsider_hip.so] [kernel-name-substring ...]
Prints per kernel: VGPR spill stores in total / under a narrowed mask.  Exit code 1 when a named kernel has any of the latter."""
import os, re, subprocess, sys, tempfile, shutil

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def disassemble(so):
    tmp = tempfile.mkdtemp()
    try:
        dst = os.path.join(tmp, "lib.so")
        shutil.copy(so, dst)
        subprocess.run([OBJDUMP, "--offloading", dst], cwd=tmp, check=True, capture_output=True)
        co = [f for f in os.listdir(tmp) if "gfx950" in f]
        if not co:
            raise RuntimeError("no gfx950 code object in " + so)
        return subprocess.run([OBJDUMP, "-d", os.path.join(tmp, co[0])], check=True, capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def scan(dis):
    """{kernel: (scratch stores, scratch stores under a narrowed exec mask, [their lines])}"""
    out = {}
    name, depth, tot, bad, lines = None, 0, 0, 0, []
    for line in dis.split("\n"):
        m = re.match(r"^[0-9a-f]{16} <(_Z[^>]+)>:", line)
        if m:
            if name:
                out[name] = (tot, bad, lines)
            name, depth, tot, bad, lines = m.group(1), 0, 0, 0, []
            continue
        t = line.strip().split("//")[0].strip()
        if not t or name is None:
            continue
        # if / else regions: `s_and_saveexec_b64` opens one (the else flip `s_andn2_saveexec` / `s_xor_b64 exec` and further
        # narrowing by `s_and_b64 exec, exec, ...` stay inside it), `s_or_b64 exec, exec, saved` closes it
        if re.match(r"s_and_saveexec_b64", t) or re.match(r"s_mov_b64 exec, s\[", t) or re.match(r"s_lshl_b64 exec,", t):
            depth += 1
        elif re.match(r"s_or_b64 exec, exec,", t):
            depth = max(depth - 1, 0)
        elif re.match(r"s_mov_b64 exec, -1", t) or re.match(r"s_endpgm", t):
            depth = 0
        if t.startswith("scratch_store"):
            tot += 1
            if depth > 0:
                bad += 1
                lines.append(t)
    if name:
        out[name] = (tot, bad, lines)
    return out


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".so") else os.path.join(here, "insider_amd", "libinsider_hip.so")
    pats = [a for a in sys.argv[1:] if not a.endswith(".so")]
    res = scan(disassemble(so))
    rc = 0
    for k in sorted(res):
        tot, bad, lines = res[k]
        if tot == 0:
            continue
        short = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().split("(")[0]
        print(f"{short}: {tot} spill stores, {bad} under a narrowed exec mask")
        if bad and any(p in k for p in pats):
            rc = 1
    sys.exit(rc)
