"""ctypes binding of libdavo_hip.so (include/davo_hip.h).

There is no CPU fallback: if the HIP library is missing or cannot be loaded this module
raises, and so does every product entry point above it."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(_HERE), "include")
# translation units of the library (csrc/ctx.h lists what each one holds); built in parallel, linked into one .so
UNITS = ("api", "forward", "plan", "weights", "launch_f32", "launch_h3", "launch_h3s", "launch_h3w", "launch_h3_generic", "launch_misc", "comm")


# -fno-slp-vectorize: with SLP vectorisation hipcc (ROCm 7.2) fuses neighbouring scalar float updates into packed
# v_pk_fma_f32 / v_pk_mul_f32 with op_sel operand selects.  One of those forms miscomputes on gfx950: `v_pk_fma_f32 ...
# op_sel:[0,1,0]` (the LOW result lane takes the HIGH register of the 64-bit src1 pair) sporadically reads the selected operand
# as 0 in lanes 48-63 once three or more waves share a SIMD beside matrix-dense neighbours - in the fused pose-head epilogue of
# cnv7 that was one forward in three off by up to 7e-3 (round-2 library included).  Established in round 4 by lane-level
# records and single-change builds (profiles/r04_flake*_variants.log): the same products without the
# select, or with the select on src0, never fail; idle cycles only thin it; the distance from the last matrix instruction, an
# in-place destination, the register addend and the packed wave reduction are all innocent.  The vectoriser cannot be told to
# avoid one operand-select form, and packed float32 beside matrix instructions buys nothing here (26.35 k vs 26.35 k
# triplets/s at B = 32, profiles/r03_noslp_ab.log; MI355X_MICROARCH.md prices it as a loss), so the library is built without
# SLP vectorisation and tools/check_isa.py fails the build on that form specifically and on any packed float32 arithmetic.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize"]


def lib_path(suffix=None):
    """DAVO_LIB_SUFFIX (or `suffix`) selects an experimental build made by tools/build_variant.py, e.g. the
    `_tuning` library with the measurement knobs compiled in; unset = the product library."""
    if suffix is None:
        suffix = os.environ.get("DAVO_LIB_SUFFIX", "")
    return os.path.join(_HERE, "libdavo_hip%s.so" % suffix)


LIB_PATH = lib_path()

EXPORTS = (
    "davo_create", "davo_load_weight", "davo_weights_missing", "davo_forward", "davo_forward_device", "davo_submit", "davo_wait", "davo_pending",
    "davo_last_error", "davo_destroy", "davo_device_malloc", "davo_device_free", "davo_memcpy_h2d",
    "davo_memcpy_d2h", "davo_synchronize", "davo_set_stream", "davo_set_inflight", "davo_profile_enable",
    "davo_profile_reset", "davo_profile_entry", "davo_profile_samples", "davo_last_plan", "davo_set_option", "davo_set_precision", "davo_set_impl", "davo_debug_read", "davo_conv2d_same",
    "davo_host_alloc", "davo_host_free", "davo_host_register", "davo_host_unregister", "davo_calibrate", "davo_activation_range", "davo_set_activation_shifts", "davo_range_stats", "davo_range_report",
    "davo_comm_preload", "davo_comm_unique_id", "davo_comm_init", "davo_comm_size", "davo_allgather_poses", "davo_allgather_poses_device",
    "davo_comm_allreduce", "davo_comm_barrier", "davo_comm_destroy", "davo_plan_layer", "davo_tile_filter_rows",
)
COMM_ID_BYTES = 128


class DavoVariant(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("cin_per_frame", "cnv6_out", "se_act", "norm_flow",
                                              "abs_mode", "att_source", "mask_rgb", "mask_info")]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))) + \
           [os.path.join(INCLUDE, "davo_hip.h")]


LLVM_OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


class PackedF32Error(RuntimeError):
    """The linked device code holds packed float32 arithmetic (see HIPCC_FLAGS above): the build is refused."""


def scan_packed_f32(so_path):
    """Disassemble every gfx950 code object of a linked library and list the packed float32 arithmetic in it:
    -> [(kernel, instruction line)].  Part of build(): a library that contains `v_pk_{fma,mul,add}_f32` is not installed, whatever
    put it there (a compiler upgrade that ignores -fno-slp-vectorize, a builtin, inline asm).  ~5 s."""
    import re
    import shutil
    import tempfile
    pat = re.compile(r"^\s*v_pk_(fma|mul|add)_f32\b")
    hits = []
    d = tempfile.mkdtemp(prefix="davo_isa_")
    try:
        tmp = os.path.join(d, "lib.so")
        shutil.copy(so_path, tmp)
        subprocess.check_call([LLVM_OBJDUMP, "--offloading", tmp], stdout=subprocess.DEVNULL, cwd=d)     # code objects land beside the input
        objs = sorted(f for f in os.listdir(d) if "amdgcn" in f)
        if not objs:
            raise PackedF32Error("no gfx950 code object found in %s: the packed-float32 scan cannot vouch for it" % so_path)
        for f in objs:
            kernel = None
            out = subprocess.run([LLVM_OBJDUMP, "-d", os.path.join(d, f)], check=True, capture_output=True, text=True).stdout
            for line in out.splitlines():
                m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
                if m:
                    kernel = m.group(1)
                elif pat.match(line):
                    hits.append((kernel, line.strip()))
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return hits


def build(force=False, verbose=False, suffix=None, extra_flags=(), allow_packed_f32=False):
    """hipcc --offload-arch=gfx950 -> davo_amd/libdavo_hip.so (in-tree, so it travels to the GPU box).
    One object per translation unit, compiled in parallel (the f16x3 instantiations dominate), then linked."""
    out = lib_path(suffix)
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(s) for s in sources()):
        return out
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = HIPCC_FLAGS + ["-fPIC", "-Wall", "-Wno-unused-function"] + \
        list(extra_flags) + os.environ.get("DAVO_EXTRA_HIPCC_FLAGS", "").split()
    objdir = os.path.join(CSRC, "build%s" % (suffix if suffix is not None else os.environ.get("DAVO_LIB_SUFFIX", "")))
    os.makedirs(objdir, exist_ok=True)
    headers = [s for s in sources() if s.endswith(".h")]
    newest_header = max(os.path.getmtime(h) for h in headers)

    def compile_unit(u):
        src, obj = os.path.join(CSRC, u + ".hip"), os.path.join(objdir, u + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), newest_header):
            return obj
        cmd = [hipcc] + flags + ["-c", "-o", obj, src]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(UNITS), os.cpu_count() or 4)) as ex:
        objs = list(ex.map(compile_unit, UNITS))
    staged = out + ".unchecked"
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", staged] + objs + ["-ldl", "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    # the guard of DESIGN.md section 4 ("A flaky sum") is part of the build, not of a test: a library with packed float32
    # arithmetic in its device code is never installed (experiment builds of tools/build_variant.py may ask for it)
    hits = [] if allow_packed_f32 else scan_packed_f32(staged)
    if hits:
        os.remove(staged)
        raise PackedF32Error("%d packed float32 instruction(s) in the device code, e.g. %s: `%s' - one operand-select form of them "
                             "miscomputes on gfx950 (DESIGN.md section 4); the library was not installed"
                             % (len(hits), hits[0][0], hits[0][1]))
    os.replace(staged, out)
    return out


_lib = None
_lib_lock = __import__("threading").Lock()


def lib():
    global _lib
    if _lib is not None:
        return _lib
    with _lib_lock:                      # start-up threads (loader pinning, RCCL preload, the main thread) may all come here first
        return _load()


def _load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libdavo_hip.so is not built (%s missing): run `python -c 'import "
                           "__graft_entry__ as g; g.build()'`; there is no CPU fallback" % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    vp, i, f32p = ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_float)
    L.davo_create.argtypes = [ctypes.POINTER(vp), i, i, i, i, ctypes.POINTER(DavoVariant)]
    L.davo_load_weight.argtypes = [vp, ctypes.c_char_p, f32p, ctypes.POINTER(ctypes.c_int64), i]
    L.davo_weights_missing.argtypes = [vp]
    L.davo_forward.argtypes = [vp, i, vp, vp, vp, vp]
    L.davo_forward_device.argtypes = [vp, i, vp, vp, vp, vp, ctypes.POINTER(ctypes.c_float)]
    L.davo_submit.argtypes = [vp, i, vp, vp, vp, vp, i]
    L.davo_wait.argtypes = [vp, i]
    L.davo_pending.argtypes = [vp]
    L.davo_last_error.argtypes = [vp]
    L.davo_last_error.restype = ctypes.c_char_p
    L.davo_destroy.argtypes = [vp]
    L.davo_destroy.restype = None
    L.davo_device_malloc.argtypes = [vp, ctypes.c_size_t, ctypes.POINTER(vp)]
    L.davo_device_free.argtypes = [vp, vp]
    L.davo_host_alloc.argtypes = [i, ctypes.c_size_t, ctypes.POINTER(vp)]
    L.davo_host_free.argtypes = [vp]
    L.davo_host_register.argtypes = [i, vp, ctypes.c_size_t]
    L.davo_host_unregister.argtypes = [vp]
    L.davo_calibrate.argtypes = [vp, i, vp, vp, vp, ctypes.POINTER(i)]
    L.davo_activation_range.argtypes = [vp, f32p, ctypes.POINTER(i), i]
    L.davo_set_activation_shifts.argtypes = [vp, ctypes.POINTER(i)]
    llp = ctypes.POINTER(ctypes.c_longlong)
    L.davo_range_stats.argtypes = [vp, llp, llp, llp]
    L.davo_range_report.argtypes = [vp]
    L.davo_range_report.restype = ctypes.c_char_p
    L.davo_memcpy_h2d.argtypes = [vp, vp, vp, ctypes.c_size_t]
    L.davo_memcpy_d2h.argtypes = [vp, vp, vp, ctypes.c_size_t]
    L.davo_synchronize.argtypes = [vp]
    L.davo_set_stream.argtypes = [vp, vp]
    L.davo_set_inflight.argtypes = [vp, i]
    L.davo_profile_enable.argtypes = [vp, i]
    L.davo_profile_reset.argtypes = [vp]
    L.davo_profile_entry.argtypes = [vp, i, ctypes.c_char_p, i, ctypes.POINTER(i), ctypes.POINTER(ctypes.c_double)]
    L.davo_profile_samples.argtypes = [vp, ctypes.c_char_p, i, f32p, i]
    L.davo_last_plan.argtypes = [vp, i, i, ctypes.POINTER(i), ctypes.POINTER(i)]
    L.davo_set_option.argtypes = [vp, ctypes.c_char_p, i]
    L.davo_set_precision.argtypes = [vp, i]
    L.davo_set_impl.argtypes = [vp, i]
    L.davo_debug_read.argtypes = [vp, ctypes.c_char_p, f32p, ctypes.c_size_t]
    L.davo_conv2d_same.argtypes = [i, f32p, i, i, i, i, f32p, i, i, f32p, i, i, i, i, f32p, ctypes.c_char_p, i]
    ip = ctypes.POINTER(i)
    L.davo_plan_layer.argtypes = [i, i, i, ip, ip, ip]
    L.davo_tile_filter_rows.argtypes = [i] * 8 + [ip, ip, i, ip]
    L.davo_comm_preload.argtypes = [ctypes.c_char_p, i]
    L.davo_comm_unique_id.argtypes = [vp, ctypes.c_char_p, i]
    L.davo_comm_init.argtypes = [vp, i, i, vp]
    L.davo_comm_size.argtypes = [vp, ip, ip]
    L.davo_allgather_poses.argtypes = [vp, f32p, i, i, f32p, f32p]
    L.davo_allgather_poses_device.argtypes = [vp, vp, i, vp, f32p]
    L.davo_comm_allreduce.argtypes = [vp, ctypes.POINTER(ctypes.c_double), i]
    L.davo_comm_barrier.argtypes = [vp]
    L.davo_comm_destroy.argtypes = [vp]
    for name in EXPORTS:
        fn = getattr(L, name)
        if name not in ("davo_last_error", "davo_destroy", "davo_range_report"):
            fn.restype = ctypes.c_int
    _lib = L
    return L
