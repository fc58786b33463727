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


sys.path.insert(0, ROOT)
from davo_amd._lib import HIPCC_FLAGS                                # noqa: E402  (the library's own device-code flags)

PACKED_F32 = re.compile(r"^\s*(v_pk_(fma|mul|add)_f32)\b")
# v_pk_*_f32 whose LOW result lane takes the HIGH register of the 64-bit src1 pair (op_sel:[x,1,x]): the one instruction form
# behind round 3's flaky pose sums (profiles/r04_flake*_variants.log)
SRC1_SELECT = re.compile(r"^\s*v_pk_(fma|mul|add)_f32\b.*\bop_sel:\[\s*[01]\s*,\s*1\b")


def compile_to_asm(out, unit="launch_h3.hip"):
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + HIPCC_FLAGS + ["-S", "--cuda-device-only",
                           "-o", out, os.path.join(ROOT, "davo_amd", "csrc", unit)], stderr=subprocess.DEVNULL)


def check_no_packed_f32(path):
    """Two properties, the second the narrow one.  (1) No packed float32 arithmetic at all: the library is built with
    -fno-slp-vectorize (davo_amd/_lib.py), because (2) cannot be asked of the vectoriser and packed f32 beside matrix instructions
    buys nothing here (26.35 k vs 26.35 k triplets/s, profiles/r03_noslp_ab.log).  (2) No v_pk_{fma,mul,add}_f32 with
    op_sel:[x,1,x]: on gfx950 that form's low lane sporadically reads its selected operand as 0 in lanes 48-63 when three or more
    waves share a SIMD beside matrix-dense neighbours (299 of 299 forwards wrong with it, 0 of 299 with the same products
    formed without the select: tools/exp/flake_count.py on the DAVO_POSE_EXP builds)."""
    hits, selects = {}, []
    kernel = None
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kernel = m.group(1)
        m = PACKED_F32.match(line)
        if m:
            hits[kernel] = hits.get(kernel, 0) + 1
        if SRC1_SELECT.match(line):
            selects.append("%s: `%s': packed float32 with a low-from-high select on src1 (miscomputes on gfx950, DESIGN.md section 4)"
                           % (kernel, line.strip()))
    return selects + ["%s: %d packed float32 instruction(s)" % (k, n) for k, n in sorted(hits.items(), key=lambda kv: str(kv[0]))]


def check_patch_loops(path):
    """The persistent patch kernels (csrc/conv_patch_h3.h) prefetch the next tile's patch by LDS-DMA while a tile computes,
    and wait for it with hand-written, counted `s_waitcnt vmcnt(n)` (inline asm).  A vmcnt wait that the COMPILER adds inside
    the tile loop (e.g. for weight registers it cannot prove loaded: an inline-asm wait is opaque to its wait-count pass)
    sits behind that DMA and silently serialises the prefetch.  Guard: inside every loop of a conv_patch kernel that holds
    an s_barrier, each vmcnt wait must be an inline-asm one."""
    problems, n_kernels = [], 0
    lines = open(path).read().splitlines()
    i = 0
    while i < len(lines):
        m = re.match(r"^(_ZN4davo18conv_patch_cnv\w+):", lines[i])
        if not m:
            i += 1
            continue
        name, j = m.group(1), i + 1
        while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
            j += 1
        body = lines[i:j]
        n_kernels += 1
        if "ILb1E" in name:
            i = j
            continue                                    # the FUSED cnv1 variant fills its patch through registers: ordinary loads
        # loop membership of every line from the compiler's block annotations ("=>This ... Loop Header", "in Loop: Header=BBx_y")
        member, cur = [], None
        for l in body:
            mm = re.match(r"^(?:\.L(BB\d+_\d+):|; %bb\.\d+:)(.*)$", l)
            if mm:
                note = mm.group(2)
                if "Loop Header" in note and "=>" in note:
                    cur = mm.group(1)
                else:
                    hh = re.search(r"in Loop: Header=(BB\d+_\d+)", note)
                    cur = hh.group(1) if hh else None
            member.append(cur)
        tile_headers = {member[k] for k, l in enumerate(body) if member[k] and l.strip().startswith("s_barrier")}
        if not tile_headers:
            problems.append("%s: no tile loop with a barrier found" % name)
        for k, l in enumerate(body):
            if not l.strip().startswith("s_waitcnt") or "vmcnt" not in l or member[k] not in tile_headers:
                continue
            prev = next((body[q].strip() for q in range(k - 1, -1, -1) if body[q].strip()), "")
            if not prev.startswith(";;#ASMSTART"):
                problems.append("%s: compiler-placed `%s' inside the tile loop" % (name, l.strip()))
        i = j
    return n_kernels, problems


KINDS = ("conv_igemm_h3", "conv_igemm_h3_mainrem", "conv_igemm_h3s", "conv_igemm_h3w", "conv_igemm_h3w64", "conv_igemm_h3w128")


def check(path, kinds=None):
    """-> (kernels checked, chunk bodies checked, problems); `kinds` (a dict) receives the kernel count per kernel family:
    conv_igemm_h3 (16x16x32 instantiations only), conv_igemm_h3_mainrem (the merged cnv5 / cnv6 grid: both tile bodies are
    16x16x32) and conv_igemm_h3s (the 208x256 tile, its own counted lgkmcnt / vmcnt waits)."""
    name, in_chunk, saw_barrier, n_chunks, n_kernels, problems, pending = None, False, False, 0, 0, [], []
    mfma_in_chunk = 0
    for line in open(path):
        m = re.match(r"^(_ZN4davo(?:13(conv_igemm_h3)I|21(conv_igemm_h3_mainrem)I|14(conv_igemm_h3s)I|14(conv_igemm_h3w)I|"
                     r"16(conv_igemm_h3w64)I|17(conv_igemm_h3w128)I)\w+):", line)
        if m:
            name, in_chunk, saw_barrier = m.group(1), False, False
            kind = next(g for g in m.groups()[1:] if g)      # (round 5: the four-wave kernels of conv_igemm_h3w.h count their waits the same way)
            # conv_igemm_h3<..., DMA=true, SMALLC, M16=true, NSTG, RATE>: only the 16x16x32 form has the hand-counted loop
            m16 = kind != "conv_igemm_h3" or re.search(r"Lb1ELb[01]ELb1ELi\dELi\dEEEvNS", name) is not None
            if not m16:
                name = None
            else:
                n_kernels += 1
                if kinds is not None:
                    kinds[kind] = kinds.get(kind, 0) + 1
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
    kinds = {}
    if len(sys.argv) > 1:
        nk, nc, problems = check(sys.argv[1], kinds)
    else:
        nk, nc, problems = 0, 0, []
        from concurrent.futures import ThreadPoolExecutor
        with tempfile.TemporaryDirectory() as d:
            units = ("launch_h3.hip", "launch_h3s.hip", "launch_h3w.hip", "launch_misc.hip", "launch_f32.hip", "launch_h3_generic.hip")
            outs = {u: os.path.join(d, u.replace(".hip", ".s")) for u in units}
            with ThreadPoolExecutor(max_workers=min(len(units), os.cpu_count() or 2)) as ex:
                list(ex.map(lambda u: compile_to_asm(outs[u], u), units))
            for unit in ("launch_h3.hip", "launch_h3s.hip", "launch_h3w.hip"):     # the merged grid lives in launch_h3.hip, the 208x256 tile and the four-wave tiles in their own units
                k, c, pr = check(outs[unit], kinds)
                nk, nc, problems = nk + k, nc + c, problems + pr
            for unit in units:
                problems += ["%s: %s" % (unit, p) for p in check_no_packed_f32(outs[unit])]
            npk, pprob = check_patch_loops(outs["launch_misc.hip"])
        for kind in KINDS:                                            # a family the name pattern no longer matches is a hole, not a pass
            if not kinds.get(kind):
                problems.append("no %s kernel found: the guard does not see that kernel family" % kind)
    print("%d hand-scheduled kernels (%s), %d chunk bodies checked, %d problem(s)" %
          (nk, ", ".join("%s: %d" % (k, kinds.get(k, 0)) for k in KINDS), nc, len(problems)))
    for p in problems[:20]:
        print("  " + p)
    if len(sys.argv) > 1:
        npk, pprob = 1, []
    else:
        print("%d patch kernels checked, %d problem(s)" % (npk, len(pprob)))
        for p in pprob[:20]:
            print("  " + p)
    return 1 if problems or pprob or nk == 0 or nc == 0 or npk == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
