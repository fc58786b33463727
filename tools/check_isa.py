#!/usr/bin/env python
"""Guard for the hand-counted LDS waits of the f16x3 matrix loop (csrc/conv_igemm_h3.h, H3_CHUNK16).

`s_waitcnt lgkmcnt(n)` releases fragments on the assumption that only in-order LDS reads are outstanding.  A scalar
memory load (returns out of order, same counter) or any other LDS/GDS/message instruction that the compiler might
one day place inside the loop would break that silently, so this script compiles the library's device code to
assembly and checks every chunk body (from the first ds_read_b128 after an s_barrier to the last matrix instruction
before the next s_barrier) of every 16x16x32 kernel.  Run by tests/test_abi.py; also: python tools/check_isa.py [file.s]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FORBIDDEN = re.compile(r"^\s*(s_load_|s_buffer_load|s_scratch_load|ds_bpermute|ds_permute|ds_swizzle|ds_write|ds_add|ds_read_(?!b128)|s_sendmsg|s_memtime|s_memrealtime|flat_)")


def compile_to_asm(out):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                           "-o", out, os.path.join(ROOT, "davo_amd", "csrc", "launch_h3.hip")], stderr=subprocess.DEVNULL)


def check(path):
    name, in_chunk, saw_barrier, n_chunks, n_kernels, problems, pending = None, False, False, 0, 0, [], []
    mfma_in_chunk = 0
    for line in open(path):
        m = re.match(r"^(_ZN4davo13conv_igemm_h3\w+):", line)
        if m:
            name, in_chunk, saw_barrier = m.group(1), False, False
            m16 = re.search(r"Lb1ELb[01]ELb1ELi\dELi\dEEEvNS", name) is not None   # <..., DMA=true, SMALLC, M16=true, NSTG, RATE>
            if not m16:
                name = None
            else:
                n_kernels += 1
            continue
        if name is None:
            continue
        if line.startswith(".Lfunc_end"):
            name = None
            continue
        ins = line.strip()
        if ins.startswith("s_barrier") or ins.startswith("s_endpgm"):
            if in_chunk:
                if mfma_in_chunk == 0:
                    problems.append("%s: a chunk body without matrix instructions" % name)
                n_chunks += 1
            in_chunk, saw_barrier, mfma_in_chunk, pending = False, True, 0, []
            continue
        if saw_barrier and not in_chunk and ins.startswith("ds_read_b128"):
            in_chunk, pending = True, []
        if in_chunk:
            if ins.startswith("v_mfma"):
                mfma_in_chunk += 1
                problems.extend(pending)              # something forbidden sat between the reads and this MFMA
                pending = []
            elif FORBIDDEN.match(line):
                pending.append("%s: `%s' inside a chunk body" % (name, ins.split(";")[0].strip()))
    return n_kernels, n_chunks, problems


def main():
    if len(sys.argv) > 1:
        nk, nc, problems = check(sys.argv[1])
    else:
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "launch_h3.s")
            compile_to_asm(out)
            nk, nc, problems = check(out)
    print("%d 16x16x32 kernels, %d chunk bodies checked, %d problem(s)" % (nk, nc, len(problems)))
    for p in problems[:20]:
        print("  " + p)
    return 1 if problems or nk == 0 or nc == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
