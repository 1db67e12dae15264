"""odr-dabmod_amd -- MI355X-native DAB COFDM hot path.

Python plumbing over the C-ABI of include/dabgpu.h (libdabgpu.so: hand-written
gfx950 HIP kernels).  PyTorch is used only for device memory, streams and
torch.distributed; all arithmetic happens in the HIP library.  There is no CPU
fallback: if the library or a gfx950 device is missing, construction raises.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_DIR, "csrc")
# DABGPU_LIB selects another build of the same library (tuning sweeps); never a fallback
LIB_PATH = os.environ.get("DABGPU_LIB") or os.path.join(_CSRC, "libdabgpu.so")

STAGE_GAIN, STAGE_FIR, STAGE_RESAMPLE, STAGE_POLY, STAGE_NOGUARD = 1, 2, 4, 8, 1 << 8
GAIN_FIX, GAIN_MAX, GAIN_VAR = 0, 1, 2

EXPORTS = [
    "dabgpu_create", "dabgpu_destroy", "dabgpu_last_error", "dabgpu_version", "dabgpu_get_geometry",
    "dabgpu_set_gain", "dabgpu_set_fir_taps", "dabgpu_set_fir_default_taps",
    "dabgpu_set_window_overlap", "dabgpu_set_resampler", "dabgpu_set_poly", "dabgpu_set_lut",
    "dabgpu_qpsk_process", "dabgpu_freq_interleave_process", "dabgpu_phase_reference_process",
    "dabgpu_diff_mod_process", "dabgpu_null_symbol_process", "dabgpu_signal_mux_process",
    "dabgpu_ofdm_process", "dabgpu_gain_process", "dabgpu_guard_process", "dabgpu_fir_process",
    "dabgpu_resampler_process", "dabgpu_poly_process", "dabgpu_chain_out_bytes_per_frame",
    "dabgpu_chain_process", "dabgpu_chain_process_dev", "dabgpu_symbols_process_dev",
    "dabgpu_synchronize",
    "dabgpu_chain_submit", "dabgpu_chain_collect", "dabgpu_set_cfr", "dabgpu_get_cfr_stats",
    "dabgpu_cic_equalizer_process", "dabgpu_set_tii", "dabgpu_tii_process",
    "dabgpu_format_size", "dabgpu_format_process", "dabgpu_format_process_dev",
    "dabgpu_set_output_format", "dabgpu_get_num_clipped", "dabgpu_fir_inverse_design",
    "dabgpu_set_fir_boundary_mode", "dabgpu_debug_last_variant", "dabgpu_debug_trace",
    "dabgpu_set_lanes", "dabgpu_wait_for_stream", "dabgpu_stream_wait_for", "dabgpu_set_handover_frames",
    "dabgpu_post_process_dev", "dabgpu_debug_lanes", "dabgpu_set_gain_rounding",
]

FORMATS = {"s16": (1, np.int16), "u8": (2, np.uint8), "s8": (3, np.int8)}


class DabGpuError(RuntimeError):
    pass


def source_hash():
    """SHA-256 (first 16 hex digits) over the sources libdabgpu.so is built from: ties a set of profiler counters
    (profiles/traffic.json) to the kernels they were collected on."""
    import hashlib
    h = hashlib.sha256()
    names = sorted(n for n in os.listdir(_CSRC) if n.endswith((".hip", ".h")) or n == "Makefile")
    for name in names:
        h.update(name.encode())
        h.update(open(os.path.join(_CSRC, name), "rb").read())
    h.update(open(os.path.join(os.path.dirname(_CSRC), "..", "include", "dabgpu.h"), "rb").read())
    return h.hexdigest()[:16]


def build(verbose=False):
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", _CSRC, "-j%d" % max(2, min(8, os.cpu_count() or 2))]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)
    return LIB_PATH


class _Config(C.Structure):
    _fields_ = [("mode", C.c_int), ("device", C.c_int), ("max_frames", C.c_int),
                ("chunks_per_frame", C.c_int)]


class _CfrStats(C.Structure):
    _fields_ = [("num_clip", C.c_uint64), ("num_error_clip", C.c_uint64), ("num_samples", C.c_uint64),
                ("mer_symbol", C.c_int), ("mer_sum_iq", C.c_double), ("mer_sum_delta", C.c_double),
                ("nb_symbols", C.c_int), ("papr_before", C.c_double * 2 * 154), ("papr_after", C.c_double * 2 * 154)]


class _Geometry(C.Structure):
    _fields_ = [("mode", C.c_int), ("nb_symbols", C.c_int), ("carriers", C.c_int),
                ("spacing", C.c_int), ("null_size", C.c_int), ("sym_size", C.c_int),
                ("tf_input_bytes", C.c_size_t), ("tf_samples", C.c_size_t)]


_lib = None


def load_library():
    """dlopen libdabgpu.so.  torch is imported first so that both share one
    libamdhip64 (the wheel bundles its own copy under the same soname)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DabGpuError("%s is missing: run __graft_entry__.build() (there is no CPU fallback)"
                          % LIB_PATH)
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is part of the image
        pass
    lib = C.CDLL(LIB_PATH)
    vp, sz, szp, u = C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_uint
    lib.dabgpu_create.argtypes = [C.POINTER(_Config), C.POINTER(vp)]
    lib.dabgpu_destroy.argtypes = [vp]
    lib.dabgpu_destroy.restype = None
    lib.dabgpu_last_error.argtypes = [vp]
    lib.dabgpu_last_error.restype = C.c_char_p
    lib.dabgpu_version.restype = C.c_char_p
    lib.dabgpu_get_geometry.argtypes = [vp, C.POINTER(_Geometry)]
    lib.dabgpu_set_gain.argtypes = [vp, C.c_int, C.c_float, C.c_float, C.c_float]
    lib.dabgpu_set_fir_taps.argtypes = [vp, C.POINTER(C.c_float), sz]
    lib.dabgpu_set_fir_default_taps.argtypes = [vp]
    lib.dabgpu_set_window_overlap.argtypes = [vp, sz]
    lib.dabgpu_set_fir_boundary_mode.argtypes = [vp, C.c_int]
    lib.dabgpu_set_gain_rounding.argtypes = [vp, C.c_int]
    lib.dabgpu_debug_last_variant.argtypes = [vp, C.c_char_p, sz]
    lib.dabgpu_debug_trace.argtypes = [vp, C.c_int]
    lib.dabgpu_set_resampler.argtypes = [vp, sz, sz]
    lib.dabgpu_set_poly.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.dabgpu_set_lut.argtypes = [vp, C.c_float, C.POINTER(C.c_float)]
    for n in ("qpsk", "freq_interleave", "ofdm", "gain", "guard", "fir", "resampler", "poly"):
        getattr(lib, "dabgpu_%s_process" % n).argtypes = [vp, vp, sz, vp, sz, szp]
    lib.dabgpu_phase_reference_process.argtypes = [vp, vp, sz, szp]
    lib.dabgpu_null_symbol_process.argtypes = [vp, vp, sz, szp]
    lib.dabgpu_diff_mod_process.argtypes = [vp, vp, sz, vp, sz, vp, sz, szp]
    lib.dabgpu_signal_mux_process.argtypes = [vp, vp, sz, vp, sz, vp, sz, szp]
    lib.dabgpu_chain_out_bytes_per_frame.argtypes = [vp, u]
    lib.dabgpu_chain_out_bytes_per_frame.restype = sz
    lib.dabgpu_chain_process.argtypes = [vp, vp, sz, u, vp, sz, szp]
    lib.dabgpu_chain_process_dev.argtypes = [vp, vp, sz, u, vp, sz, szp, vp]
    lib.dabgpu_symbols_process_dev.argtypes = [vp, vp, sz, u, vp, sz, szp, vp]
    lib.dabgpu_chain_submit.argtypes = [vp, vp, sz, u]
    lib.dabgpu_chain_collect.argtypes = [vp, C.POINTER(vp), szp]
    lib.dabgpu_synchronize.argtypes = [vp]
    lib.dabgpu_set_cfr.argtypes = [vp, C.c_int, C.c_float, C.c_float]
    lib.dabgpu_get_cfr_stats.argtypes = [vp, sz, C.POINTER(_CfrStats)]
    lib.dabgpu_cic_equalizer_process.argtypes = [vp, sz, C.c_int, vp, sz, vp, sz, szp]
    lib.dabgpu_set_tii.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.dabgpu_tii_process.argtypes = [vp, vp, sz, vp, sz, szp]
    lib.dabgpu_format_size.argtypes = [C.c_int]
    lib.dabgpu_format_size.restype = sz
    lib.dabgpu_format_process.argtypes = [vp, vp, sz, C.c_int, vp, sz, szp, szp]
    lib.dabgpu_format_process_dev.argtypes = [vp, vp, sz, C.c_int, vp, sz, szp, vp, vp]
    lib.dabgpu_set_output_format.argtypes = [vp, C.c_int]
    lib.dabgpu_get_num_clipped.argtypes = [vp, szp]
    lib.dabgpu_set_lanes.argtypes = [vp, C.c_int]
    lib.dabgpu_set_handover_frames.argtypes = [vp, C.c_int]
    lib.dabgpu_debug_lanes.argtypes = [vp, C.POINTER(C.c_int)]
    lib.dabgpu_wait_for_stream.argtypes = [vp, vp]
    lib.dabgpu_stream_wait_for.argtypes = [vp, vp]
    lib.dabgpu_post_process_dev.argtypes = [vp, vp, sz, u, vp, sz, szp, vp]
    lib.dabgpu_fir_inverse_design.argtypes = [C.POINTER(C.c_float), sz, C.POINTER(C.c_float), C.POINTER(C.c_double)]
    _lib = lib
    return lib


def fir_inverse_design(taps):
    """Host-side helper (no device): the 160-tap inverse of a FIR of up to 45 taps on the occupied carriers of a Mode I symbol,
    as the frame kernel's equalised-boundary variant uses it.  Returns (ok, g, fit)."""
    lib = load_library()
    taps = np.ascontiguousarray(taps, np.float32)
    g = np.zeros(160, np.float32)
    fit = C.c_double()
    rc = lib.dabgpu_fir_inverse_design(_f32p(taps), taps.size, _f32p(g), C.byref(fit))
    return rc == 0, g, fit.value


def _f32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class Modulator:
    """One device context: geometry tables, settings and scratch for one stream.

    Method names follow the reference plugins (src/DabModulator.cpp:385-419);
    errors the reference throws as std::runtime_error surface as DabGpuError
    with the same message.
    """

    def __init__(self, mode=1, device=0, max_frames=1, chunks_per_frame=0):
        self._lib = load_library()
        cfg = _Config(mode, device, max_frames, chunks_per_frame)
        h = C.c_void_p()
        rc = self._lib.dabgpu_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise DabGpuError("dabgpu_create: " + self._lib.dabgpu_last_error(None).decode())
        self._h = h
        g = _Geometry()
        self._lib.dabgpu_get_geometry(self._h, C.byref(g))
        self.geometry = {n: getattr(g, n) for n, _ in _Geometry._fields_}
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            self._lib.dabgpu_destroy(self._h)
            self._h = None

    __del__ = close

    def _chk(self, rc):
        if rc != 0:
            raise DabGpuError(self._lib.dabgpu_last_error(self._h).decode())

    # ---- settings ----------------------------------------------------------
    def set_gain(self, mode=GAIN_VAR, digital=1.0, normalise=1.0, var_variance=4.0):
        self._chk(self._lib.dabgpu_set_gain(self._h, mode, digital, normalise, var_variance))

    def set_fir_taps(self, taps=None):
        if taps is None:
            self._chk(self._lib.dabgpu_set_fir_default_taps(self._h))
        else:
            t = np.ascontiguousarray(taps, np.float32)
            self._chk(self._lib.dabgpu_set_fir_taps(self._h, _f32p(t), t.size))

    def set_window_overlap(self, overlap):
        self._chk(self._lib.dabgpu_set_window_overlap(self._h, overlap))

    def set_resampler(self, in_rate, out_rate):
        self._chk(self._lib.dabgpu_set_resampler(self._h, in_rate, out_rate))

    def set_poly(self, am, pm):
        a = np.ascontiguousarray(am, np.float32)
        p = np.ascontiguousarray(pm, np.float32)
        assert a.size == 5 and p.size == 5
        self._chk(self._lib.dabgpu_set_poly(self._h, _f32p(a), _f32p(p)))

    def set_lut(self, scalefactor, lut):
        t = np.ascontiguousarray(lut, np.float32)
        assert t.size == 32
        self._chk(self._lib.dabgpu_set_lut(self._h, float(scalefactor), _f32p(t)))

    # ---- per-stage, host arrays -------------------------------------------
    def _stage(self, name, x, out_bytes, dtype=np.complex64):
        x = np.ascontiguousarray(x)
        out = np.empty(max(out_bytes, 1), np.uint8)
        n = C.c_size_t()
        fn = getattr(self._lib, "dabgpu_%s_process" % name)
        self._chk(fn(self._h, x.ctypes.data, x.nbytes, out.ctypes.data, out_bytes, C.byref(n)))
        return out[:n.value].view(dtype)

    def qpsk(self, bits):
        bits = np.ascontiguousarray(bits, np.uint8)
        return self._stage("qpsk", bits, bits.size * 32)

    def freq_interleave(self, x):
        x = np.ascontiguousarray(x, np.complex64)
        return self._stage("freq_interleave", x, x.nbytes)

    def phase_reference(self):
        nbytes = self.geometry["carriers"] * 8
        out = np.empty(nbytes, np.uint8)
        n = C.c_size_t()
        self._chk(self._lib.dabgpu_phase_reference_process(self._h, out.ctypes.data, nbytes, C.byref(n)))
        return out[:n.value].view(np.complex64)

    def null_symbol(self):
        nbytes = self.geometry["carriers"] * 8
        out = np.empty(nbytes, np.uint8)
        n = C.c_size_t()
        self._chk(self._lib.dabgpu_null_symbol_process(self._h, out.ctypes.data, nbytes, C.byref(n)))
        return out[:n.value].view(np.complex64)

    def diff_mod(self, phase, data):
        phase = np.ascontiguousarray(phase, np.complex64)
        data = np.ascontiguousarray(data, np.complex64)
        nbytes = phase.nbytes + data.nbytes
        out = np.empty(max(nbytes, 1), np.uint8)
        n = C.c_size_t()
        self._chk(self._lib.dabgpu_diff_mod_process(self._h, phase.ctypes.data, phase.nbytes,
                                                    data.ctypes.data, data.nbytes,
                                                    out.ctypes.data, nbytes, C.byref(n)))
        return out[:n.value].view(np.complex64)

    def signal_mux(self, first, rest):
        first = np.ascontiguousarray(first, np.complex64)
        rest = np.ascontiguousarray(rest, np.complex64)
        nbytes = first.nbytes + rest.nbytes
        out = np.empty(max(nbytes, 1), np.uint8)
        n = C.c_size_t()
        self._chk(self._lib.dabgpu_signal_mux_process(self._h, first.ctypes.data, first.nbytes,
                                                      rest.ctypes.data, rest.nbytes,
                                                      out.ctypes.data, nbytes, C.byref(n)))
        return out[:n.value].view(np.complex64)

    def ofdm(self, x):
        x = np.ascontiguousarray(x, np.complex64)
        g = self.geometry
        return self._stage("ofdm", x, (g["nb_symbols"] + 1) * g["spacing"] * 8)

    def gain(self, x):
        x = np.ascontiguousarray(x, np.complex64)
        return self._stage("gain", x, x.nbytes)

    def guard(self, x):
        x = np.ascontiguousarray(x, np.complex64)
        return self._stage("guard", x, self.geometry["tf_samples"] * 8)

    def fir(self, x):
        x = np.ascontiguousarray(x, np.complex64)
        return self._stage("fir", x, x.nbytes)

    def resample(self, x, ratio_hint=8):
        x = np.ascontiguousarray(x, np.complex64)
        return self._stage("resampler", x, x.nbytes * ratio_hint)

    def poly(self, x):
        x = np.ascontiguousarray(x, np.complex64)
        return self._stage("poly", x, x.nbytes)

    def set_cfr(self, enable, clip=1.0, error_clip=1.0):
        """OfdmGenerator RC parameters cfr / clip / errorclip (src/OfdmGenerator.cpp:376-404)."""
        self._chk(self._lib.dabgpu_set_cfr(self._h, int(enable), clip, error_clip))

    def cfr_stats(self, frame=0):
        """Raw CFR statistics of frame `frame` of the most recent call with CFR on."""
        st = _CfrStats()
        self._chk(self._lib.dabgpu_get_cfr_stats(self._h, frame, C.byref(st)))
        n = st.nb_symbols
        d = {k: getattr(st, k) for k in ("num_clip", "num_error_clip", "num_samples", "mer_symbol",
                                         "mer_sum_iq", "mer_sum_delta", "nb_symbols")}
        d["papr_before"] = np.array([[st.papr_before[i][0], st.papr_before[i][1]] for i in range(n)])
        d["papr_after"] = np.array([[st.papr_after[i][0], st.papr_after[i][1]] for i in range(n)])
        return d

    def cic_equalizer(self, x, spacing, R):
        """CicEqualizer(carriers, spacing, R)::process."""
        x = np.ascontiguousarray(x, np.complex64)
        out = np.empty_like(x)
        n = C.c_size_t()
        self._chk(self._lib.dabgpu_cic_equalizer_process(self._h, spacing, R, x.ctypes.data, x.nbytes,
                                                         out.ctypes.data, out.nbytes, C.byref(n)))
        return out

    def set_tii(self, enable, comb=0, pattern=0, old_variant=False):
        self._chk(self._lib.dabgpu_set_tii(self._h, int(enable), comb, pattern, int(old_variant)))

    def tii(self, phase):
        """TII::process: phase reference symbol -> TII symbol (or zeros on idle calls)."""
        x = np.ascontiguousarray(phase, np.complex64)
        return self._stage("tii", x, x.nbytes)

    def format_convert(self, x, fmt):
        """FormatConverter (float input): returns (integer array, clipped components).
        An unknown format raises like the reference (src/FormatConverter.cpp:171-173)."""
        x = np.ascontiguousarray(x).view(np.float32).ravel()
        code, dt = FORMATS.get(fmt, (0, np.uint8))
        out = np.empty(x.size, dt)
        ob, nc = C.c_size_t(), C.c_size_t()
        self._chk(self._lib.dabgpu_format_process(self._h, x.ctypes.data, x.nbytes, code, out.ctypes.data,
                                                  out.nbytes, C.byref(ob), C.byref(nc)))
        return out[:ob.value // out.itemsize], int(nc.value)

    def format_convert_dev(self, d_in, fmt, d_out, d_clipped=None, stream=None):
        """Device path: d_in complex64/float32 tensor -> d_out integer tensor (asynchronous);
        d_clipped (int64 tensor of one element, optional) is incremented."""
        code = FORMATS.get(fmt, (0, None))[0]
        n = d_in.numel() * (2 if d_in.is_complex() else 1)
        ob = C.c_size_t()
        s = self._stream_handle(d_in, stream)
        self._chk(self._lib.dabgpu_format_process_dev(
            self._h, d_in.data_ptr(), n, code, d_out.data_ptr(), d_out.numel() * d_out.element_size(),
            C.byref(ob), d_clipped.data_ptr() if d_clipped is not None else None, s))
        if not s:
            self.synchronize()
        return ob.value

    # ---- fused chain -------------------------------------------------------
    def set_fir_boundary_mode(self, direct):
        """False (default): boundary outputs of the fused FIRFilter through the taps' inverse where one exists;
        True: always the direct sum over the unfiltered samples (packed dual transform)."""
        self._chk(self._lib.dabgpu_set_fir_boundary_mode(self._h, 1 if direct else 0))

    def set_gain_rounding(self, reference):
        """False (default): gain mode var from the exact variance inside the frame kernel; True: chain calls replay the
        reference's running fp32 recurrence (src/GainControl.cpp:251-340) -- its scalars bit for bit, separate kernels."""
        self._chk(self._lib.dabgpu_set_gain_rounding(self._h, 1 if reference else 0))

    def trace(self, enable=True):
        """Turn the launch trace behind last_variant() on or off (off by default)."""
        self._chk(self._lib.dabgpu_debug_trace(self._h, 1 if enable else 0))

    def last_variant(self):
        """The kernels the most recent chain call launched (dabgpu_debug_last_variant), as a list of names."""
        buf = C.create_string_buffer(4096)
        self._chk(self._lib.dabgpu_debug_last_variant(self._h, buf, len(buf)))
        return [k for k in buf.value.decode().split("; ") if k]

    def set_output_format(self, fmt=None):
        """FormatConverter as the chain's last step: None / "complexf", or "s16" / "u8" / "s8"."""
        code = 0 if fmt in (None, "complexf") else FORMATS.get(fmt, (99, None))[0]
        self._chk(self._lib.dabgpu_set_output_format(self._h, code))
        self._out_dtype = np.complex64 if code == 0 else FORMATS[fmt][1]

    def num_clipped(self):
        """Clipped components of the most recent chain call (FormatConverter::get_num_clipped_samples)."""
        n = C.c_size_t()
        self._chk(self._lib.dabgpu_get_num_clipped(self._h, C.byref(n)))
        return int(n.value)

    def out_bytes_per_frame(self, stages):
        return self._lib.dabgpu_chain_out_bytes_per_frame(self._h, stages)

    def out_samples_per_frame(self, stages):
        """Complex samples per frame (8 bytes each as complexf; 4 / 2 with an integer output format)."""
        dt = np.dtype(getattr(self, "_out_dtype", np.complex64))
        per_sample = 8 if dt == np.complex64 else 2 * dt.itemsize
        return self._lib.dabgpu_chain_out_bytes_per_frame(self._h, stages) // per_sample

    def chain(self, bits, stages, out=None):
        """Host path: bits (n_frames x tf_input_bytes uint8) -> complex64 (n_frames x samples), or the integer
        components (n_frames x 2 * samples) when an output format is set.  `out`: a buffer to reuse (what a ModPlugin's
        Buffer is: allocated once, Buffer::setLength only grows) -- a fresh 100 MB numpy array per call is 25 000 page
        faults inside the copy."""
        bits = np.ascontiguousarray(bits, np.uint8).reshape(-1)
        per = self.geometry["tf_input_bytes"]
        if bits.size % per:
            raise DabGpuError("chain: input size not valid")
        n = bits.size // per
        dt = np.dtype(getattr(self, "_out_dtype", np.complex64))
        per_out = self.out_bytes_per_frame(stages) // dt.itemsize
        if out is None:
            out = np.empty(n * per_out, dt)
        # (checked BEFORE any reshape: reshaping a non-contiguous array makes a copy, and the caller's buffer would stay empty)
        if not isinstance(out, np.ndarray) or out.dtype != dt or out.size != n * per_out or not out.flags.c_contiguous:
            raise DabGpuError("chain: output buffer does not match (dtype %s, %d elements, C-contiguous)" % (dt, n * per_out))
        flat = out.reshape(-1)
        assert flat.size == 0 or np.shares_memory(flat, out)
        out = flat
        ob = C.c_size_t()
        self._chk(self._lib.dabgpu_chain_process(self._h, bits.ctypes.data, n, stages,
                                                 out.ctypes.data, out.nbytes, C.byref(ob)))
        return out.reshape(n, per_out)

    def submit(self, bits, stages):
        """Asynchronous host path: queue a batch (at most two in flight)."""
        bits = np.ascontiguousarray(bits, np.uint8).reshape(-1)
        per = self.geometry["tf_input_bytes"]
        if bits.size % per:
            raise DabGpuError("chain: input size not valid")
        self._chk(self._lib.dabgpu_chain_submit(self._h, bits.ctypes.data, bits.size // per, stages))

    def collect(self, copy=True):
        """Wait for the oldest batch: complex64 samples (a view of the context's pinned buffer when
        copy=False: valid until the second next submit)."""
        p, n = C.c_void_p(), C.c_size_t()
        self._chk(self._lib.dabgpu_chain_collect(self._h, C.byref(p), C.byref(n)))
        dt = np.dtype(getattr(self, "_out_dtype", np.complex64))
        a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n.value,)).view(dt)
        return a.copy() if copy else a

    def _stream_handle(self, tensor, stream):
        """HIP stream handle to launch on.  A real torch stream is used as is
        (fully asynchronous).  torch's legacy default stream has handle 0, which
        the C-ABI reads as "the context's own stream": in that case order the two
        by hand (drain torch's stream now, the context's stream after the call)."""
        import torch
        h = torch.cuda.current_stream(tensor.device).cuda_stream if stream is None else stream
        if not h:
            torch.cuda.current_stream(tensor.device).synchronize()
        return h

    def chain_dev(self, d_bits, n_frames, stages, d_out, stream=None):
        """Device path on torch tensors (coded bits -> IQ)."""
        s = self._stream_handle(d_bits, stream)
        ob = C.c_size_t()
        self._chk(self._lib.dabgpu_chain_process_dev(
            self._h, d_bits.data_ptr(), n_frames, stages, d_out.data_ptr(),
            d_out.numel() * d_out.element_size(), C.byref(ob), s))
        if not s:
            self.synchronize()
        return ob.value

    def symbols_dev(self, d_carriers, n_frames, stages, d_out, stream=None):
        """Device path on torch tensors (SignalMultiplexer output -> IQ)."""
        s = self._stream_handle(d_carriers, stream)
        ob = C.c_size_t()
        self._chk(self._lib.dabgpu_symbols_process_dev(
            self._h, d_carriers.data_ptr(), n_frames, stages, d_out.data_ptr(),
            d_out.numel() * d_out.element_size(), C.byref(ob), s))
        if not s:
            self.synchronize()
        return ob.value

    def synchronize(self):
        """Wait for everything the context has queued, on every lane."""
        self._chk(self._lib.dabgpu_synchronize(self._h))

    # ---- batches in flight inside the context (PipelinedModCodec's idiom, src/ModPlugin.cpp:90-154) ----
    def set_lanes(self, lanes):
        """Internal HIP streams that calls on the context's own stream rotate over (1 ... 4, default 3)."""
        self._chk(self._lib.dabgpu_set_lanes(self._h, int(lanes)))

    def lanes_info(self):
        """(lanes created so far, [lane i has a hardware queue of its own])."""
        m = C.c_int()
        n = self._lib.dabgpu_debug_lanes(self._h, C.byref(m))
        if n < 0:
            raise DabGpuError(self._lib.dabgpu_last_error(self._h).decode())
        return n, [bool(m.value >> i & 1) for i in range(n)]

    def set_handover_frames(self, frames):
        """FIRFilter -> Resampler hand-over in pieces of `frames` frames through a cache-resident ring (0: one piece)."""
        self._chk(self._lib.dabgpu_set_handover_frames(self._h, int(frames)))

    def wait_for_stream(self, stream):
        """What the context queues from now on starts after what the HIP stream (handle) holds now."""
        self._chk(self._lib.dabgpu_wait_for_stream(self._h, stream))

    def stream_wait_for(self, stream):
        """What is queued on the HIP stream (handle) from now on starts after everything the context has queued."""
        self._chk(self._lib.dabgpu_stream_wait_for(self._h, stream))

    def chain_dev_queued(self, d_in, n_frames, stages, d_out, from_bits=True):
        """Device path on the context's OWN stream (the lanes): returns at once; the output is complete after
        synchronize() or, in stream order, after stream_wait_for()."""
        ob = C.c_size_t()
        fn = self._lib.dabgpu_chain_process_dev if from_bits else self._lib.dabgpu_symbols_process_dev
        self._chk(fn(self._h, d_in.data_ptr(), n_frames, stages, d_out.data_ptr(),
                     d_out.numel() * d_out.element_size(), C.byref(ob), None))
        return ob.value

    def post_process_dev_queued(self, d_native, stages, d_out):
        """dabgpu_post_process_dev on the context's OWN stream (stream argument NULL), returning at once: ordered behind every
        chain call queued on the context before it, whichever lane that call went to."""
        ob = C.c_size_t()
        self._chk(self._lib.dabgpu_post_process_dev(self._h, d_native.data_ptr(), d_native.numel(), stages,
                                                    d_out.data_ptr(), d_out.numel() * d_out.element_size(), C.byref(ob), None))
        return ob.value

    def format_convert_dev_queued(self, d_in, fmt, d_out):
        """dabgpu_format_process_dev on the context's OWN stream, returning at once (same ordering as above)."""
        code = FORMATS.get(fmt, (0, None))[0]
        n = d_in.numel() * (2 if d_in.is_complex() else 1)
        ob = C.c_size_t()
        self._chk(self._lib.dabgpu_format_process_dev(self._h, d_in.data_ptr(), n, code, d_out.data_ptr(),
                                                      d_out.numel() * d_out.element_size(), C.byref(ob), None, None))
        return ob.value

    def post_process_dev(self, d_native, stages, d_out, stream=None):
        """cifRes -> cifPoly on a native-rate stream in device memory (stages: STAGE_RESAMPLE and / or STAGE_POLY)."""
        s = self._stream_handle(d_native, stream)
        ob = C.c_size_t()
        self._chk(self._lib.dabgpu_post_process_dev(self._h, d_native.data_ptr(), d_native.numel(), stages,
                                                    d_out.data_ptr(), d_out.numel() * d_out.element_size(), C.byref(ob), s))
        if not s:
            self.synchronize()
        return ob.value
