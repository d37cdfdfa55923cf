"""Static check of the shipped gfx950 code: which buffer instructions of the descriptor-addressed kernels (k_cg1b, all forms) carry a
SCALAR offset register?  (profiles/r02/k_cg1b_soffset_hazard.txt: with the row offset in the instruction's scalar offset the
fp32-p build read x from wrong rows now and then; the kernel source passes 0 everywhere since -- this makes sure the compiler does
not reintroduce one on its own, and prints what surrounds such an instruction if it does.)
usage: python tools/check_isa.py [path/to/libopenimpala_amd.so]   -> exit code 1 if any is found"""
import os, re, shutil, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def device_disassembly(lib):
    tmp = tempfile.mkdtemp(prefix="oi_isa_")
    try:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(lib, so)
        subprocess.run([OBJDUMP, "--offloading", so], check=True, capture_output=True, cwd=tmp)
        out = []
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" in f:
                out.append(subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout)
        return "\n".join(out)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def scan(text, pattern="k_cg1b"):
    """{function: [(line number in function, instruction text, context)]} of buffer instructions whose soffset is a register"""
    found, counts = {}, {}
    fn, body = None, []
    def flush():
        if fn is None or pattern not in fn:
            return
        n = 0
        for i, ins in enumerate(body):
            m = re.match(r"\s*(buffer_\w+)\s+(.*)", ins)
            if not m:
                continue
            n += 1
            ops = [o.strip() for o in m.group(2).split(",")]
            # vdata, vaddr, srsrc, soffset [modifiers]   (stores and loads alike; `off` replaces vaddr when there is none)
            soff = ops[3].split()[0] if len(ops) > 3 else "?"
            if soff not in ("0",):
                found.setdefault(fn, []).append((i, ins.strip(), [b.strip() for b in body[max(0, i - 3):i + 6]]))
        counts[fn] = n
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            flush()
            fn, body = m.group(1), []
        elif fn is not None:
            body.append(line.split("//")[0])
    flush()
    return found, counts


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "openimpala_amd", "libopenimpala_amd.so")
    found, counts = scan(device_disassembly(lib))
    print(f"{len(counts)} k_cg1b instantiations, {sum(counts.values())} buffer instructions")
    for fn, hits in found.items():
        print(f"{fn}: {len(hits)} buffer instructions with a scalar offset register")
        for i, ins, ctx in hits[:3]:
            print("   ", ins)
            for c in ctx:
                print("        |", c)
    sys.exit(1 if found else 0)
