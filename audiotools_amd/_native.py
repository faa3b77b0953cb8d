"""ctypes binding of the C-ABI HIP library ``libaudiotools_amd.so``.

The library is the product: every hot-path method of :class:`AudioSignal`
on a HIP tensor goes through one of the ``at_*`` entry points declared in
``include/audiotools_amd.h``.  There is deliberately NO CPU fallback here:
if the shared object is missing or an entry point fails, a ``RuntimeError``
is raised (``NativeError``).
"""
import ctypes
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")
# The shipped library has no run-time switches.  AT_DEV_KNOBS=1 (set by the A/B tools under tools/, never by the
# package) selects the development build of the same sources: -DAT_DEV_KNOBS=1 compiles the AT_* environment switches and
# the measurement-only kernels in, as lib/libaudiotools_amd_dev.so (`python -m audiotools_amd._native --dev` builds it).
DEV_KNOBS = os.environ.get("AT_DEV_KNOBS") == "1"
SHIP_LIB_PATH = os.path.join(LIB_DIR, "libaudiotools_amd.so")
DEV_LIB_PATH = os.path.join(LIB_DIR, "libaudiotools_amd_dev.so")
LIB_PATH = os.environ.get("AT_LIB_PATH") or (DEV_LIB_PATH if DEV_KNOBS else SHIP_LIB_PATH)
CSRC_DIR = os.path.join(_HERE, "csrc")
SOURCES = ["stft.hip", "istft.hip", "loudness.hip", "fir.hip", "firfft.hip", "irtools.hip", "specedit.hip", "fftconv.hip", "vocoder.hip", "stft_generic.hip", "longconv.hip", "resample_f16.hip"]

_lib = None
_lock = threading.Lock()


class NativeError(RuntimeError):
    """Raised when the HIP library is unavailable or an entry point fails."""


HEADERS = ["at_common.h", "fft_wave.h", "generic_fft.h"]
# per-source extra flags.  The SLP vectoriser turns the float2 butterflies into v_pk_*_f32, which issue
# at half rate on gfx950 (no throughput gain) and cost v_mov's to build the operand pairs; scalar
# code is shorter AND needs fewer registers.  Measured, same box, packed -> scalar:
# stft+mel 2.484 -> 2.406 ms, fir_fft 0.982 -> 0.817 ms, istft 2.283 -> 2.221 ms; loudness unchanged.
_NO_SLP = ["-fno-slp-vectorize"]
# loudness.hip (round 3): the scalar build of kweight_hop_energy_dma needs 86 VGPRs instead of 113 and its loop has
# fewer issue slots (803 scalar FP + 74 moves vs 605 scalar + 126 half-rate packed + 121 moves).
# istft.hip: the inverse kernels are one long dependent chain per wave at two waves per SIMD; LLVM's "max-ilp" strategy
# orders it for latency instead of register pressure (no spills, 237 -> 254 registers: the occupancy is the same) and
# measures 2-3 % faster; every other source measured slower or equal with it (profiles/r03_notes.md, s81).
_MAX_ILP = ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]
FILE_FLAGS = {"stft.hip": _NO_SLP, "istft.hip": _NO_SLP + _MAX_ILP, "firfft.hip": _NO_SLP, "longconv.hip": _NO_SLP, "loudness.hip": _NO_SLP,
              "stft_generic.hip": _NO_SLP}
LINK_FLAGS = ["-L/opt/rocm/lib", "-lrocfft", "-Wl,-rpath,/opt/rocm/lib"]


def _hipcc():
    return os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _dev_flags(dev):
    """Extra compile flags of a development build (AT_HIPCC_FLAGS etc. are honoured there only)."""
    if not dev:
        return "", "", "", ""
    return (os.environ.get("AT_HIPCC_FLAGS", ""), os.environ.get("AT_STFT_SLP", ""), os.environ.get("AT_NOSLP_ALL", ""),
            os.environ.get("AT_MAXILP_FILES", ""))


def compile_command(src, obj, dev=False):
    hipcc_flags, stft_slp, noslp_all, maxilp_files = _dev_flags(dev)
    extra = (["-DAT_DEV_KNOBS=1"] if dev else []) + hipcc_flags.split()
    per_file = FILE_FLAGS.get(os.path.basename(src), [])
    if stft_slp == "1" and os.path.basename(src) == "stft.hip":
        per_file = []
    if noslp_all == "1":          # A/B builds: every source without SLP packing
        per_file = ["-fno-slp-vectorize"]
    if os.path.basename(src) in maxilp_files.split(","):   # A/B builds of the scheduler strategy
        per_file = [f for f in per_file if f not in _MAX_ILP] + _MAX_ILP
    return [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c"] + per_file + extra + ["-o", obj, src]


def link_command(objs, out=LIB_PATH):
    return [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + list(objs) + LINK_FLAGS


def hipcc_command(out=SHIP_LIB_PATH):
    """The build as shell commands (INTEGRATION.md quotes these)."""
    objs = [os.path.join(LIB_DIR, "obj", s.replace(".hip", ".o")) for s in SOURCES]
    cmds = [compile_command(os.path.join(CSRC_DIR, s), o) for s, o in zip(SOURCES, objs)]
    return cmds + [link_command(objs, out)]


def build(force: bool = False, verbose: bool = False, out: str = None, dev: bool = None) -> str:
    """Compile every HIP source for gfx950 (one object per source, in parallel; hipcc cross-compiles
    without a GPU) and link ``lib/libaudiotools_amd.so`` (``dev=True``: the development build with the A/B
    switches, ``lib/libaudiotools_amd_dev.so``)."""
    if out is None and os.environ.get("AT_LIB_PATH"):
        out = os.environ["AT_LIB_PATH"]        # what lib() will load: a development build when the process opted in
        if dev is None:
            dev = DEV_KNOBS
    if dev is None:
        dev = out is None and DEV_KNOBS
    out = out or (DEV_LIB_PATH if dev else SHIP_LIB_PATH)
    obj_dir = os.path.join(os.path.dirname(out), "obj" if out == SHIP_LIB_PATH else "obj_" + os.path.basename(out))
    os.makedirs(obj_dir, exist_ok=True)
    hdrs = [os.path.join(CSRC_DIR, h) for h in HEADERS]
    hdr_time = max(os.path.getmtime(h) for h in hdrs)
    flag_stamp = os.path.join(obj_dir, "flags.txt")
    flags_now = repr((dev,) + _dev_flags(dev) + (FILE_FLAGS,))
    flags_same = os.path.exists(flag_stamp) and open(flag_stamp).read() == flags_now
    jobs, objs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC_DIR, s)
        obj = os.path.join(obj_dir, s.replace(".hip", ".o"))
        objs.append(obj)
        stale = (force or not flags_same or not os.path.exists(obj)
                 or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_time))
        if stale:
            cmd = compile_command(src, obj, dev)
            if verbose:
                print(" ".join(cmd))
            jobs.append((cmd, subprocess.Popen(cmd)))
    for cmd, proc in jobs:
        if proc.wait() != 0:
            raise subprocess.CalledProcessError(proc.returncode, cmd)
    if jobs or not os.path.exists(out) or any(os.path.getmtime(out) < os.path.getmtime(o) for o in objs):
        cmd = link_command(objs, out)
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        with open(flag_stamp, "w") as f:
            f.write(flags_now)
    return out


_i64, _i32, _p, _f32, _f64 = (ctypes.c_int64, ctypes.c_int, ctypes.c_void_p,
                              ctypes.c_float, ctypes.c_double)

# name -> (restype, argtypes); mirrors include/audiotools_amd.h
SIGNATURES = {
    "at_stft_twiddles_host": (_i32, [_i32, _p]),
    "at_stft_native_supported": (_i32, [_i32]),
    "at_stft_fused_supported": (_i32, [_i32]),
    "at_mel_units_host": (_i32, [_p, _i32, _i32, _p, _p]),
    "at_mel_bands_host": (_i32, [_p, _i32, _i32, _p, _p]),
    "at_stft_mel_f32": (_i32, [_p, _i64, _i64, _p, _p, _i32, _i32, _i32, _i32, _i32,
                               _i32, _i64, _p, _p, _p, _i32, _i32, _p, _p]),
    "at_istft_workspace_bytes": (_i64, [_i64, _i64, _i32, _i32]),
    "at_istft_f32": (_i32, [_p, _i64, _i64, _p, _p, _i32, _i32, _i32, _i64, _i64, _p, _p, _i64, _p]),
    "at_istft_edit_f32": (_i32, [_p, _i64, _i64, _p, _p, _i32, _i32, _i32, _i64, _i64, _p, _p, _i64,
                                 _i32, _i64, _p, _p, _p, _p, _p, _f32, _f32, _f32, _i32, _f32, _p]),
    "at_stft_adjoint_f32": (_i32, [_p, _i64, _i64, _p, _p, _i32, _i32, _p, _i64, _p]),
    "at_stft_mel_adjoint_f32": (_i32, [_p, _p, _p, _p, _i32, _i64, _i64, _p, _p, _i32, _i32, _p, _i64, _p]),
    "at_fir_per_item_f32": (_i32, [_p, _i64, _i64, _i64, _p, _i32, _i32, _i32, _i32, _p, _p]),
    "at_sinc_taps_f32": (_i32, [_p, _i64, _f32, _i32, _i32, _p, _p]),
    "at_eq_taps_f32": (_i32, [_p, _p, _i64, _i32, _i32, _i32, _i32, _p, _p]),
    "at_fir_fft_f32": (_i32, [_p, _i64, _i64, _i64, _p, _i32, _i32, _i32, _i32, _p, _p, _p]),
    "at_spec_mask_f32": (_i32, [_p, _p, _i64, _i64, _i64, _i64, _i32, _p, _p, _p, _f32, _f32, _p]),
    "at_spec_phase_shift_f32": (_i32, [_p, _p, _i64, _i64, _i64, _i64, _p, _p]),
    "at_spec_polar_elem_f32": (_i32, [_p, _p, _i64, _i64, _i64, _i64, _p, _p, _i32, _p]),
    "at_spec_maxpow_f32": (_i32, [_p, _i64, _p, _p]),
    "at_spec_gate_f32": (_i32, [_p, _p, _i64, _i64, _i64, _i64, _p, _i32, _p, _p, _i32, _p, _i32, _p]),
    "at_spec_mask_lowmag_f32": (_i32, [_p, _p, _i64, _i64, _i64, _i64, _p, _p, _f32, _i32, _f32, _p]),
    "at_phase_vocoder_frames": (_i64, [_i64, _i64, _i64]),
    "at_phase_vocoder_f32": (_i32, [_p, _i64, _i64, _i64, _i64, _i64, _i32, _p, _i64, _p]),
    "at_absmax_f32": (_i32, [_p, _i64, _i64, _p, _p, _p]),
    "at_roll_pad_f32": (_i32, [_p, _i64, _i64, _p, _i64, _p, _p]),
    "at_collect_windows_f32": (_i32, [_p, _i64, _i64, _i32, _i32, _p, _p]),
    "at_quantize_f32": (_i32, [_p, _i64, _i64, _p, _i32, _p, _p]),
    "at_overlap_add_f32": (_i32, [_p, _i64, _i64, _i32, _i32, _i64, _i64, _p, _p]),
    "at_alter_drr_f32": (_i32, [_p, _i64, _i64, _i64, _i32, _p, _p, _p]),
    "at_alter_drr_peak_f32": (_i32, [_p, _i64, _i64, _i64, _i32, _p, _p, _p, _p, _p]),
    "at_resample_f32": (_i32, [_p, _i64, _i64, _p, _p, _i32, _i32, _i32, _i32, _i32, _p, _i64, _p]),
    "at_resample_mfma_supported": (_i32, [_i32, _i32]),
    "at_resample_mfma_f32": (_i32, [_p, _i64, _i64, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _p, _i64, _p]),
    "at_resample_f16s_supported": (_i32, [_i32, _i32]),
    "at_resample_f16s_f32": (_i32, [_p, _i64, _i64, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p, _i64, _p]),
    "at_fftconv_workspace_bytes": (_i64, [_i64, _i64, _i64, _i64]),
    "at_fftconv_circ_f32": (_i32, [_p, _p, _p, _i64, _i64, _i64, _i64, _p, _p, _i64, _p]),
    "at_longconv_supported": (_i32, [_i64]),
    "at_longconv_plan": (_i32, [_i64, _p, _p]),
    "at_longconv_table_floats": (_i64, [_i64]),
    "at_longconv_tables_host": (_i32, [_i64, _p, _i64]),
    "at_longconv_workspace_bytes": (_i64, [_i64, _i64, _i64, _i64]),
    "at_longconv_circ_f32": (_i32, [_p, _p, _p, _i64, _i64, _i64, _i64, _p, _p, _p, _i64, _p]),
    "at_longconv_room_f32": (_i32, [_p, _p, _i64, _i64, _p, _p, _i64, _i64, _i64, _i64, _p, _p, _p, _p, _p, _i64, _p]),
    "at_lufs_workspace_bytes": (_i64, [_i64, _i64, _i64, _i32, _i32]),
    "at_lufs_f32": (_i32, [_p, _i64, _i64, _i64, _p, _p, _i32, _i32, _i32, _f64, _f32,
                           _i32, _p, _p, _i64, _p]),
}


# measurement-only entry points of the development build (-DAT_DEV_KNOBS=1): never part of the shipped ABI
DEV_SIGNATURES = {
    # zero-compute twin of at_stft_mel_f32 for the n_fft 2048 / hop 512 kernel: same arguments, grid, runs, addresses, load /
    # store instructions and cache policy, no transform (bench.py: roofline.floor_ms; tools/regime.py)
    "at_stft_mel_floor_f32": (_i32, [_p, _i64, _i64, _p, _p, _i32, _i32, _i32, _i32, _i32,
                                     _i32, _i64, _p, _p, _p, _i32, _i32, _p, _p]),
}

_dev_lib = None


def dev_lib(build_if_missing: bool = False):
    """ctypes handle of the DEVELOPMENT library with its measurement entry points bound, loaded next to the product library
    (its own handle: nothing of it is reachable through `lib()`); None when it has not been built."""
    global _dev_lib
    if _dev_lib is not None:
        return _dev_lib
    with _lock:
        if _dev_lib is None:
            if not os.path.exists(DEV_LIB_PATH):
                if not build_if_missing:
                    return None
                build(dev=True)
            handle = lib() if LIB_PATH == DEV_LIB_PATH else ctypes.CDLL(DEV_LIB_PATH)
            for name, (res, args) in DEV_SIGNATURES.items():
                fn = getattr(handle, name)
                fn.restype = res
                fn.argtypes = args
            _dev_lib = handle
    return _dev_lib


def lib():
    """Load (once) and return the ctypes handle.  Fails loudly."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            how = ("`python -m audiotools_amd._native --dev` (the development build AT_DEV_KNOBS=1 asks for)" if LIB_PATH == DEV_LIB_PATH
                   else "`python -c 'import __graft_entry__ as g; g.build()'`")
            raise NativeError(f"{LIB_PATH} not found: build it with {how} (hipcc --offload-arch=gfx950). There is no CPU fallback.")
        try:
            handle = ctypes.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover
            raise NativeError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError as e:
                raise NativeError(f"{LIB_PATH} does not export {name}") from e
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


class NativeUnsupported(NativeError):
    """AT_ERR_UNSUPPORTED: a valid request this entry point has no kernel for (callers that own another native route to the
    same result may catch it; everything else treats it as the NativeError it is)."""


def check(code: int, what: str):
    if code == 0:
        return
    if code == -1:
        raise NativeError(f"{what}: invalid argument")
    if code == -2:
        raise NativeUnsupported(f"{what}: unsupported configuration")
    raise NativeError(f"{what}: HIP error {-(code) - 1000}")


def ptr(t):
    """Raw device/host pointer of a tensor (or None)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def current_stream(device):
    import torch

    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


if __name__ == "__main__":      # python -m audiotools_amd._native [--dev] [--force]
    import sys

    print(build(force="--force" in sys.argv, verbose=True, dev="--dev" in sys.argv))
