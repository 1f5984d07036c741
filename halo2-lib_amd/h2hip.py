"""ctypes binding of libh2hip's C ABI (include/h2hip.h).  Host arrays are numpy uint64: field elements
(n,4), affine points (n,8), Jacobian points (n,12), all Montgomery limbs exactly as halo2curves stores them."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
POINT_JACOBIAN = 0
POINT_AFFINE = 1
PERM_FIRST, PERM_LAST, PERM_CHAIN, PERM_PRODUCT = 1, 2, 4, 8   # H2HIP_PERM_* term mask of quotient_permutation_set
BASES_PLAIN = 0
BASES_PRECOMPUTE = 1

_vp, _sz, _u32, _int = C.c_void_p, C.c_size_t, C.c_uint32, C.c_int
# names accepted by h2hip_profile_get for the stages of one MSM
MSM_PROFILE_NAMES = ("msm_digits_kernel", "msm_hist_kernel", "msm_hist_scan_kernel", "scan_kernels", "msm_scatter_kernel", "msm_accum_kernel",
                     "msm_merge_kernel", "msm_presum_kernel", "msm_seg_kernel", "msm_winsum_kernel", "msm_fold_kernel")
_PROTOS = {
    "h2hip_last_error": (C.c_char_p, []),
    "h2hip_version": (_int, []),
    "h2hip_device_count": (_int, [C.POINTER(_int)]),
    "h2hip_init": (_int, [_int, _vp, C.POINTER(_vp)]),
    "h2hip_destroy": (None, [_vp]),
    "h2hip_sync": (_int, [_vp]),
    "h2hip_set_param": (_int, [_vp, C.c_char_p, _int]),
    "h2hip_get_param": (_int, [_vp, C.c_char_p, C.POINTER(_int)]),
    "h2hip_malloc": (_int, [_vp, _sz, C.POINTER(_vp)]),
    "h2hip_free": (_int, [_vp, _vp]),
    "h2hip_upload": (_int, [_vp, _vp, _vp, _sz]),
    "h2hip_download": (_int, [_vp, _vp, _vp, _sz]),
    "h2hip_host_register": (_int, [_vp, _vp, _sz]),
    "h2hip_host_unregister": (_int, [_vp, _vp]),
    "h2hip_profile_enable": (_int, [_vp, _int]),
    "h2hip_profile_reset": (_int, [_vp]),
    "h2hip_profile_get_busy": (_int, [_vp, C.c_char_p, C.POINTER(C.c_double)]),
    "h2hip_profile_get": (_int, [_vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "h2hip_profile_filter": (_int, [_vp, C.c_char_p]),
    "h2hip_profile_dump": (_int, [_vp, C.c_char_p, _sz, C.POINTER(_sz)]),
    "h2hip_timer_start": (_int, [_vp]),
    "h2hip_timer_stop": (_int, [_vp, C.POINTER(C.c_double)]),
    "h2hip_bases_upload": (_int, [_vp, _vp, _sz, _u32, C.POINTER(_vp)]),
    "h2hip_bases_from_device": (_int, [_vp, _vp, _sz, _u32, C.POINTER(_vp)]),
    "h2hip_bases_free": (None, [_vp, _vp]),
    "h2hip_bases_len": (_sz, [_vp]),
    "h2hip_msm_g1": (_int, [_vp, _vp, _vp, _sz, _int, _vp]),
    "h2hip_msm_g1_dev": (_int, [_vp, _vp, _vp, _sz, _int, _vp]),
    "h2hip_msm_g1_batch": (_int, [_vp, _vp, C.POINTER(_vp), _sz, _sz, _int, _vp]),
    "h2hip_msm_g1_batch_dev": (_int, [_vp, _vp, C.POINTER(_vp), _sz, _sz, _int, _vp]),
    "h2hip_msm_g1_multi_dev": (_int, [_vp, C.POINTER(_vp), C.POINTER(_vp), _sz, _sz, _int, _vp]),
    "h2hip_msm_g2": (_int, [_vp, _vp, _vp, _sz, _vp]),
    "h2hip_msm_g2_dev": (_int, [_vp, _vp, _vp, _sz, _vp]),
    "h2hip_g1_to_lagrange": (_int, [_vp, _vp, _u32, _u32, C.POINTER(_vp)]),
    "h2hip_params_kzg_setup": (_int, [_vp, _u32, _vp, _u32, C.POINTER(_vp), C.POINTER(_vp)]),
    "h2hip_g1_fixed_base_mul_batch_dev": (_int, [_vp, _vp, _vp, _sz, _vp]),
    "h2hip_bases_download": (_int, [_vp, _vp, _vp]),
    "h2hip_g1_validate_dev": (_int, [_vp, _vp, _sz, C.POINTER(_sz)]),
    "h2hip_g1_decompress_batch_dev": (_int, [_vp, _vp, _sz, _vp, _u32, _u32]),
    "h2hip_g1_sum_jacobian_dev": (_int, [_vp, _vp, _sz, _int, _vp]),
    "h2hip_g1_sum_partials_host": (_int, [_vp, _sz, _sz, _int, _vp]),
    "h2hip_best_fft": (_int, [_vp, _vp, _vp, _u32]),
    "h2hip_best_fft_dev": (_int, [_vp, _vp, _vp, _u32]),
    "h2hip_ifft": (_int, [_vp, _vp, _vp, _u32, _vp]),
    "h2hip_ifft_dev": (_int, [_vp, _vp, _vp, _u32, _vp]),
    "h2hip_coeff_to_extended": (_int, [_vp, _vp, _u32, _vp, _u32, _vp, _vp]),
    "h2hip_coeff_to_extended_dev": (_int, [_vp, _vp, _u32, _vp, _u32, _vp, _vp]),
    "h2hip_extended_to_coeff": (_int, [_vp, _vp, _u32, _vp, _vp, _vp]),
    "h2hip_extended_to_coeff_dev": (_int, [_vp, _vp, _u32, _vp, _vp, _vp]),
    "h2hip_fr_add_batch_dev": (_int, [_vp, _vp, _vp, _vp, _sz]),
    "h2hip_fr_sub_batch_dev": (_int, [_vp, _vp, _vp, _vp, _sz]),
    "h2hip_fr_mul_batch_dev": (_int, [_vp, _vp, _vp, _vp, _sz]),
    "h2hip_fr_mul_add_batch_dev": (_int, [_vp, _vp, _vp, _vp, _vp, _sz]),
    "h2hip_fr_batch_invert_dev": (_int, [_vp, _vp, _sz]),
    "h2hip_fr_prefix_product_dev": (_int, [_vp, _vp, _vp, _sz]),
    "h2hip_fr_grand_product_dev": (_int, [_vp, _vp, _vp, _vp, _sz]),
    "h2hip_fr_grand_products_dev": (_int, [_vp, _vp, _vp, _vp, _sz, _sz, _int]),
    "h2hip_fr_eval_polynomial_dev": (_int, [_vp, _vp, _sz, _vp, _vp]),
    "h2hip_fr_eval_polynomial_batch_dev": (_int, [_vp, C.POINTER(_vp), C.POINTER(_sz), _vp, _sz, _vp]),
    "h2hip_fr_kate_division_dev": (_int, [_vp, _vp, _vp, _sz, _vp]),
    "h2hip_fr_kate_division_multi_dev": (_int, [_vp, _vp, _vp, _sz, _vp, _vp, _u32]),
    "h2hip_fr_kate_division_multi_acc_dev": (_int, [_vp, _vp, _vp, _sz, _vp, _vp, _u32]),
    "h2hip_fr_kate_division_sets_dev": (_int, [_vp, _vp, _vp, _sz, _vp, _vp, C.POINTER(_u32), _sz, _int]),
    "h2hip_fr_kate_division_range_dev": (_int, [_vp, _vp, _vp, _sz, _vp, _vp, _vp, _u32]),
    "h2hip_quotient_flex_gate_dev": (_int, [_vp, _vp, _vp, _vp, _u32, _u32, _vp]),
    "h2hip_quotient_lookup_dev": (_int, [_vp] * 10 + [_u32, _u32, _vp, _vp, _vp]),
    "h2hip_quotient_permutation_set_dev": (_int, [_vp, _vp, _vp, _vp, C.POINTER(_vp), C.POINTER(_vp), _u32, _u32, _vp, _vp, _vp, _u32, _u32, _u32,
                                                  C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "h2hip_fr_axpy_dev": (_int, [_vp, _vp, _vp, _vp, _sz]),
    "h2hip_fr_scale_dev": (_int, [_vp, _vp, _vp, _sz]),
    "h2hip_fr_axpby_dev": (_int, [_vp, _vp, _vp, _vp, _vp, _sz]),
    "h2hip_quotient_flex_gate_batch_dev": (_int, [_vp, _vp, _vp, _vp, _sz, _u32, _u32, _vp]),
    "h2hip_quotient_lookups_dev": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp, _vp, _u32, _u32, _vp, _vp, _vp]),
    "h2hip_quotient_permutation_sets_dev": (_int, [_vp, _vp, _vp, _u32, _vp, _vp, _u32, _u32, _vp, _vp, _vp, _u32, _u32, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "h2hip_permutation_product_terms_sets_dev": (_int, [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _sz, _vp, _vp, _vp, _vp]),
    "h2hip_permutation_product_terms_rows_dev": (_int, [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _sz, _sz, _vp, _vp, _vp, _vp]),
    "h2hip_lookup_permute_presorted_batch_dev": (_int, [_vp, _vp, _vp, _sz, _vp, _vp, _sz]),
    "h2hip_array_rng_fill": (None, [_vp, _vp, _sz]),
    "h2hip_rng_seed_from_u64": (None, [C.c_uint64, _vp]),
    "h2hip_chacha_rng_init": (None, [_vp, _vp, _int]),
    "h2hip_chacha_block": (None, [_vp, C.c_uint64, C.c_uint64, _int, _vp]),
    "h2hip_chacha_rng_fill": (None, [_vp, _vp, _sz]),
    "h2hip_rng_chacha_fill_dev": (_int, [_vp, _vp, _sz, _vp, _int, C.c_uint64]),
    "h2hip_ifft_batch_dev": (_int, [_vp, _vp, _sz, _vp, _u32, _vp]),
    "h2hip_coeff_to_extended_batch_dev": (_int, [_vp, _vp, _u32, _vp, _u32, _sz, _vp, _vp]),
    "h2hip_fr_linear_combination_dev": (_int, [_vp, _vp, _vp, _vp, _sz, _sz]),
    "h2hip_fr_sub_low_dev": (_int, [_vp, _vp, _vp, _u32]),
    "h2hip_assigned_resolve_dev": (_int, [_vp, _vp, _vp, _vp, _sz]),
    "h2hip_permutation_product_terms_dev": (_int, [_vp, _vp, _vp, C.POINTER(_vp), C.POINTER(_vp), _u32, _u32, _sz, _vp, _vp, _vp, _vp]),
    "h2hip_lookup_product_terms_dev": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    "h2hip_plonk_shape_of": (_int, [_vp, _vp]),
    "h2hip_plonk_keygen": (_int, [_vp, _vp, _vp, _vp, C.POINTER(_vp), _vp, _sz, C.POINTER(_vp)]),
    "h2hip_plonk_pk_free": (None, [_vp, _vp]),
    "h2hip_plonk_pk_commitments": (_int, [_vp, _vp, _vp]),
    "h2hip_plonk_pk_set_transcript_repr": (_int, [_vp, _vp]),
    "h2hip_plonk_pk_set_sharding": (_int, [_vp, _vp, _vp, _vp, _sz, _sz, _u32]),
    "h2hip_plonk_pk_last_exchanges": (_int, [_vp, _vp, _sz, _vp]),
    "h2hip_comm_rccl_unique_id": (_int, [_vp]),
    "h2hip_comm_init_rccl": (_int, [_vp, _vp, _int, _int, C.POINTER(_vp)]),
    "h2hip_comm_init_callback": (_int, [_int, _int, _vp, _vp, C.POINTER(_vp)]),
    "h2hip_comm_info": (_int, [_vp, C.POINTER(_int), C.POINTER(_int), C.POINTER(_int)]),
    "h2hip_comm_destroy": (None, [_vp]),
    "h2hip_comm_allgather_dev": (_int, [_vp, _vp, _vp, _sz, _vp]),
    "h2hip_comm_alltoall_dev": (_int, [_vp, _vp, _vp, _sz, _vp]),
    "h2hip_comm_allgather_host": (_int, [_vp, _vp, _vp, _sz, _vp]),
    "h2hip_fr_coset_scale_batch_dev": (_int, [_vp, C.POINTER(_vp), C.POINTER(_vp), _sz, _sz, _vp]),
    "h2hip_fr_coset_gather_dev": (_int, [_vp, _vp, _vp, C.POINTER(_u32), _u32, _u32, _sz]),
    "h2hip_fr_coset_interleave_dev": (_int, [_vp, _vp, _vp, C.POINTER(_u32), _u32, _sz]),
    "h2hip_fr_coset_combine_dev": (_int, [_vp, _vp, _vp, C.POINTER(_u32), _u32, _sz, _vp, _vp]),
    "h2hip_plonk_stage_name": (C.c_char_p, [_int]),
    "h2hip_plonk_create_proof": (_int, [_vp, _vp, C.POINTER(_vp), _int, C.POINTER(_vp), C.POINTER(_sz), _vp, _vp, _vp, _sz, C.POINTER(_sz),
                                        C.POINTER(C.c_double)]),
    "h2hip_divide_by_vanishing_poly_dev": (_int, [_vp, _vp, _u32, _u32, _vp, _vp]),
    "h2hip_lookup_permute_dev": (_int, [_vp, _vp, _vp, _sz, _vp, _vp]),
    "h2hip_lookup_sorted_table_bytes": (_sz, [_sz]),
    "h2hip_lookup_table_sort_dev": (_int, [_vp, _vp, _sz, _vp]),
    "h2hip_lookup_permute_presorted_dev": (_int, [_vp, _vp, _vp, _sz, _vp, _vp]),
    "h2hip_poseidon_set_spec": (_int, [_vp, _u32, _u32, _u32, _vp, _vp]),
    "h2hip_poseidon_permute_batch_dev": (_int, [_vp, _vp, _vp, _u32, _sz]),
    "h2hip_plonk_verify_proof": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(_vp), C.POINTER(_sz), _vp, _sz, C.POINTER(_int)]),
    "h2hip_pairing_check": (_int, [_vp, _vp, _sz, C.POINTER(_int)]),
    "h2hip_blake2b": (_int, [_vp, C.c_uint, _vp, _sz, _vp]),
    "h2hip_bench_gather": (_int, [_vp, _u32, _sz, _u32, _u32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "h2hip_bench_modmul29": (_int, [_vp, _u32, _u32, _u32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "h2hip_bench_modmul": (_int, [_vp, _u32, _u32, _u32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
}
# symbols added by later translation units register themselves here (see fr_ops section below)
EXPORTED_SYMBOLS = list(_PROTOS)


class H2HipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libh2hip error {code}: {msg}")
        self.code = code


def library_path() -> str:
    return os.path.join(_HERE, "csrc", "libh2hip.so")


_LIBS = {}


_TORCH_HIP_PRELOADED = False


def _initialise_torch_hip_runtime_first():
    """PyTorch's ROCm wheels bundle their own libamdhip64.so / libhsa-runtime64.so (unversioned SONAMEs), libh2hip.so links the system
    ROCm's (libamdhip64.so.7): a process that uses both holds two HSA runtimes, and the one that initialises SECOND must not be torch's
    (measured on the GPU box: libh2hip first, then torch.cuda -> "No HIP GPUs are available"; the other order works).  So when a torch
    installation is present, its HIP runtime is loaded and initialised here, before libh2hip.so is opened — without importing torch."""
    global _TORCH_HIP_PRELOADED
    if _TORCH_HIP_PRELOADED:
        return
    _TORCH_HIP_PRELOADED = True
    try:
        import importlib.util

        spec = importlib.util.find_spec("torch")
        if not spec or not spec.origin:
            return
        lib = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if not os.path.exists(lib):
            return
        n = C.c_int(0)
        C.CDLL(lib, mode=C.RTLD_GLOBAL).hipGetDeviceCount(C.byref(n))
    except Exception:   # no torch, or a CPU-only box: nothing to order
        pass


def load_library(path: Optional[str] = None):
    """dlopen libh2hip.so and attach prototypes.  Fails loudly when the HIP extension has not been built."""
    path = os.path.abspath(path or library_path())
    if path in _LIBS:
        return _LIBS[path]
    if os.path.basename(path) == "libh2hip.so":
        _initialise_torch_hip_runtime_first()
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'); "
            "halo2-lib_amd has no CPU fallback")
    lib = C.CDLL(path)
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)   # AttributeError here = the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _LIBS[path] = lib
    return lib


def _fe(a) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a.reshape(-1, 4)


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(_vp)


class Bases:
    """Resident G1Affine bases (an SRS column: ParamsKZG.g or .g_lagrange)."""

    def __init__(self, ctx: "Context", handle: int, n: int):
        self.ctx, self.handle, self.n = ctx, handle, n

    def free(self):
        if self.handle:
            self.ctx.lib.h2hip_bases_free(self.ctx.handle, self.handle)
            self.handle = None

    def __len__(self):
        return self.n


class Context:
    """One per GPU/process; serialises work on one HIP stream (optionally a caller-provided hipStream_t,
    e.g. torch.cuda.current_stream().cuda_stream)."""

    def __init__(self, device: int = 0, stream: Optional[int] = None, lib_path: Optional[str] = None):
        self.lib = load_library(lib_path)
        h = _vp()
        self.handle = None
        self._chk(self.lib.h2hip_init(device, _vp(stream) if stream else None, C.byref(h)))
        self.handle = h
        self.device = device

    # -- plumbing
    def _chk(self, rc: int):
        if rc != 0:
            raise H2HipError(rc, self.lib.h2hip_last_error().decode(errors="replace"))

    def close(self):
        if self.handle:
            self.lib.h2hip_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        self._chk(self.lib.h2hip_sync(self.handle))

    def set_param(self, name: str, value: int):
        self._chk(self.lib.h2hip_set_param(self.handle, name.encode(), int(value)))

    def get_param(self, name: str) -> int:
        v = _int()
        self._chk(self.lib.h2hip_get_param(self.handle, name.encode(), C.byref(v)))
        return v.value

    # -- device memory
    def malloc(self, nbytes: int) -> int:
        p = _vp()
        self._chk(self.lib.h2hip_malloc(self.handle, nbytes, C.byref(p)))
        return p.value

    def free(self, dptr: int):
        self._chk(self.lib.h2hip_free(self.handle, _vp(dptr)))

    def upload(self, dptr: int, host: np.ndarray):
        host = np.ascontiguousarray(host)
        self._chk(self.lib.h2hip_upload(self.handle, _vp(dptr), _ptr(host), host.nbytes))

    def host_register(self, arr: np.ndarray):
        """page-lock a numpy array that will be uploaded repeatedly (keep the array alive until host_unregister)"""
        self._chk(self.lib.h2hip_host_register(self.handle, _ptr(arr), arr.nbytes))

    def host_unregister(self, arr: np.ndarray):
        self._chk(self.lib.h2hip_host_unregister(self.handle, _ptr(arr)))

    def to_device(self, host: np.ndarray) -> int:
        host = np.ascontiguousarray(host)
        d = self.malloc(host.nbytes)
        self.upload(d, host)
        return d

    def download(self, dptr: int, shape, dtype=np.uint64) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        self._chk(self.lib.h2hip_download(self.handle, _ptr(out), _vp(dptr), out.nbytes))
        return out

    # -- profiling / timing
    def profile_enable(self, on: bool = True):
        self._chk(self.lib.h2hip_profile_enable(self.handle, 1 if on else 0))

    def profile_filter(self, prefix: str = ""):
        self._chk(self.lib.h2hip_profile_filter(self.handle, prefix.encode()))

    def profile_reset(self):
        self._chk(self.lib.h2hip_profile_reset(self.handle))

    def profile_get_busy(self, prefix: str) -> float:
        """ms during which at least one launch of the kernels matching `prefix` was running (union of the launch spans)"""
        ms = C.c_double(0)
        self._chk(self.lib.h2hip_profile_get_busy(self.handle, prefix.encode(), C.byref(ms)))
        return ms.value

    def profile_get(self, prefix: str):
        ms, cnt = C.c_double(), C.c_uint64()
        self._chk(self.lib.h2hip_profile_get(self.handle, prefix.encode(), C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value

    def profile_dump(self) -> dict:
        """{kernel name: (total_ms, launches, busy_ms)} since the last profile_reset"""
        need = _sz(0)
        self._chk(self.lib.h2hip_profile_dump(self.handle, None, 0, C.byref(need)))
        buf = C.create_string_buffer(need.value + 16)
        self._chk(self.lib.h2hip_profile_dump(self.handle, buf, len(buf), None))
        out = {}
        for line in buf.value.decode().splitlines():
            name, ms, cnt, busy = line.split()
            out[name] = (float(ms), int(cnt), float(busy))
        return out

    def timer_start(self):
        self._chk(self.lib.h2hip_timer_start(self.handle))

    def timer_stop(self) -> float:
        ms = C.c_double()
        self._chk(self.lib.h2hip_timer_stop(self.handle, C.byref(ms)))
        return ms.value

    # -- MSM (arithmetic::best_multiexp)
    def bases_upload(self, points: np.ndarray, flags: int = BASES_PLAIN) -> Bases:
        pts = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 8)
        h = _vp()
        self._chk(self.lib.h2hip_bases_upload(self.handle, _ptr(pts), len(pts), flags, C.byref(h)))
        return Bases(self, h, len(pts))

    def bases_from_device(self, dptr: int, n: int, flags: int = BASES_PLAIN) -> Bases:
        h = _vp()
        self._chk(self.lib.h2hip_bases_from_device(self.handle, _vp(dptr), n, flags, C.byref(h)))
        return Bases(self, h, n)

    def msm(self, bases: Bases, scalars: np.ndarray, point_format: int = POINT_AFFINE) -> np.ndarray:
        s = _fe(scalars)
        out = np.zeros((1, 8 if point_format == POINT_AFFINE else 12), dtype=np.uint64)
        self._chk(self.lib.h2hip_msm_g1(self.handle, bases.handle, _ptr(s), len(s), point_format, _ptr(out)))
        return out

    def msm_dev(self, bases: Bases, scalars_dptr: int, n: int, point_format: int = POINT_JACOBIAN) -> np.ndarray:
        out = np.zeros((1, 8 if point_format == POINT_AFFINE else 12), dtype=np.uint64)
        self._chk(self.lib.h2hip_msm_g1_dev(self.handle, bases.handle, _vp(scalars_dptr), n, point_format, _ptr(out)))
        return out

    def g1_to_lagrange(self, g: "Bases", k: int, flags: int = 0) -> "Bases":
        """g_to_lagrange: Lagrange bases from the first 2^k monomial bases (group inverse FFT on the device)"""
        out = _vp()
        self._chk(self.lib.h2hip_g1_to_lagrange(self.handle, g.handle, k, flags, C.byref(out)))
        return Bases(self, out, 1 << k)

    def params_kzg_setup(self, k: int, s: np.ndarray, flags: int = BASES_PLAIN):
        """(g, g_lagrange) resident base sets of ParamsKZG::setup(k) for the toxic-waste scalar s"""
        g, gl = _vp(), _vp()
        self._chk(self.lib.h2hip_params_kzg_setup(self.handle, k, _ptr(_fe(s)), flags, C.byref(g), C.byref(gl)))
        return Bases(self, g, 1 << k), Bases(self, gl, 1 << k)

    def g1_fixed_base_mul(self, base: np.ndarray, scalars: np.ndarray) -> np.ndarray:
        s = _fe(scalars)
        b = np.ascontiguousarray(base, dtype=np.uint64).reshape(1, 8)
        ds, do = self.to_device(s), self.malloc(64 * max(len(s), 1))
        try:
            self._chk(self.lib.h2hip_g1_fixed_base_mul_batch_dev(self.handle, _ptr(b), _vp(ds), len(s), _vp(do)))
            return self.download(do, (len(s), 8))
        finally:
            self.free(ds)
            self.free(do)

    def bases_download(self, bases: Bases) -> np.ndarray:
        out = np.empty((bases.n, 8), dtype=np.uint64)
        self._chk(self.lib.h2hip_bases_download(self.handle, bases.handle, _ptr(out)))
        return out

    def g1_sum_jacobian_dev(self, points_dptr: int, n: int, point_format: int = POINT_JACOBIAN) -> np.ndarray:
        out = np.zeros((1, 8 if point_format == POINT_AFFINE else 12), dtype=np.uint64)
        self._chk(self.lib.h2hip_g1_sum_jacobian_dev(self.handle, _vp(points_dptr), n, point_format, _ptr(out)))
        return out

    def msm_batch_dev(self, bases: Bases, scalar_dptrs, n: int, point_format: int = POINT_JACOBIAN) -> np.ndarray:
        """several independent MSMs over the same bases, pipelined over the lanes of the context; returns (count, 8|12)"""
        count = len(scalar_dptrs)
        arr = (_vp * count)(*[_vp(int(p)) for p in scalar_dptrs])
        out = np.zeros((count, 8 if point_format == POINT_AFFINE else 12), dtype=np.uint64)
        self._chk(self.lib.h2hip_msm_g1_batch_dev(self.handle, bases.handle, arr, n, count, point_format, _ptr(out)))
        return out

    def msm_multi_dev(self, bases_per_column, scalar_dptrs, n: int, point_format: int = POINT_JACOBIAN) -> np.ndarray:
        """one pipelined call for columns over different base sets (a Bases per column)"""
        count = len(scalar_dptrs)
        assert len(bases_per_column) == count
        barr = (_vp * max(count, 1))(*[b.handle for b in bases_per_column])
        arr = (_vp * max(count, 1))(*[_vp(int(p)) for p in scalar_dptrs])
        out = np.zeros((count, 8 if point_format == POINT_AFFINE else 12), dtype=np.uint64)
        self._chk(self.lib.h2hip_msm_g1_multi_dev(self.handle, barr, arr, n, count, point_format, _ptr(out)))
        return out

    def msm_g2(self, points: np.ndarray, scalars: np.ndarray) -> np.ndarray:
        """sum_i scalars[i] * points[i] over G2; points (n, 16) u64 affine Montgomery (x.c0, x.c1, y.c0, y.c1), returns (1, 16)"""
        pts = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 16)
        s = _fe(scalars)
        assert len(pts) == len(s)
        out = np.zeros((1, 16), dtype=np.uint64)
        self._chk(self.lib.h2hip_msm_g2(self.handle, _ptr(pts), _ptr(s), len(s), _ptr(out)))
        return out

    def msm_batch(self, bases: Bases, scalar_columns, point_format: int = POINT_JACOBIAN) -> np.ndarray:
        """the same with the scalar columns in host memory (equal lengths): column j+1 is uploaded while column j computes"""
        cols = [_fe(c) for c in scalar_columns]
        count, n = len(cols), (len(cols[0]) if cols else 0)
        assert all(len(c) == n for c in cols)
        arr = (_vp * count)(*[_vp(c.ctypes.data) for c in cols])
        out = np.zeros((count, 8 if point_format == POINT_AFFINE else 12), dtype=np.uint64)
        self._chk(self.lib.h2hip_msm_g1_batch(self.handle, bases.handle, arr, n, count, point_format, _ptr(out)))
        return out

    # -- NTT family (arithmetic::best_fft, EvaluationDomain::*)
    def best_fft(self, a: np.ndarray, omega: np.ndarray, log_n: int) -> np.ndarray:
        a = _fe(a).copy()
        assert len(a) == 1 << log_n
        self._chk(self.lib.h2hip_best_fft(self.handle, _ptr(a), _ptr(_fe(omega)), log_n))
        return a

    def best_fft_dev(self, dptr: int, omega: np.ndarray, log_n: int):
        self._chk(self.lib.h2hip_best_fft_dev(self.handle, _vp(dptr), _ptr(_fe(omega)), log_n))

    def ifft(self, a: np.ndarray, omega_inv: np.ndarray, log_n: int, divisor: np.ndarray) -> np.ndarray:
        a = _fe(a).copy()
        self._chk(self.lib.h2hip_ifft(self.handle, _ptr(a), _ptr(_fe(omega_inv)), log_n, _ptr(_fe(divisor))))
        return a

    def ifft_dev(self, dptr: int, omega_inv: np.ndarray, log_n: int, divisor: np.ndarray):
        self._chk(self.lib.h2hip_ifft_dev(self.handle, _vp(dptr), _ptr(_fe(omega_inv)), log_n, _ptr(_fe(divisor))))

    def ifft_batch_dev(self, dptrs, omega_inv: np.ndarray, log_n: int, divisor: np.ndarray):
        """ifft_dev over several resident columns at once (32 columns per launch)"""
        arr = (_vp * max(len(dptrs), 1))(*[_vp(d) for d in dptrs])
        self._chk(self.lib.h2hip_ifft_batch_dev(self.handle, arr, len(dptrs), _ptr(_fe(omega_inv)), log_n, _ptr(_fe(divisor))))

    def coeff_to_extended_batch_dev(self, coeffs_dptrs, k: int, out_dptrs, ext_k: int, ext_omega: np.ndarray, zeta: np.ndarray):
        """coeff_to_extended_dev over several resident columns at once"""
        assert len(coeffs_dptrs) == len(out_dptrs)
        a = (_vp * max(len(coeffs_dptrs), 1))(*[_vp(d) for d in coeffs_dptrs])
        o = (_vp * max(len(out_dptrs), 1))(*[_vp(d) for d in out_dptrs])
        self._chk(self.lib.h2hip_coeff_to_extended_batch_dev(self.handle, a, k, o, ext_k, len(coeffs_dptrs), _ptr(_fe(ext_omega)), _ptr(_fe(zeta))))

    def coeff_to_extended(self, coeffs: np.ndarray, k: int, ext_k: int, ext_omega: np.ndarray, zeta: np.ndarray) -> np.ndarray:
        a = _fe(coeffs)
        assert len(a) == 1 << k
        out = np.empty((1 << ext_k, 4), dtype=np.uint64)
        self._chk(self.lib.h2hip_coeff_to_extended(self.handle, _ptr(a), k, _ptr(out), ext_k, _ptr(_fe(ext_omega)), _ptr(_fe(zeta))))
        return out

    def coeff_to_extended_dev(self, coeffs_dptr: int, k: int, out_dptr: int, ext_k: int, ext_omega: np.ndarray, zeta: np.ndarray):
        self._chk(self.lib.h2hip_coeff_to_extended_dev(self.handle, _vp(coeffs_dptr), k, _vp(out_dptr), ext_k, _ptr(_fe(ext_omega)), _ptr(_fe(zeta))))

    def extended_to_coeff(self, a: np.ndarray, ext_k: int, ext_omega_inv: np.ndarray, ext_divisor: np.ndarray, zeta_inv: np.ndarray) -> np.ndarray:
        a = _fe(a).copy()
        self._chk(self.lib.h2hip_extended_to_coeff(self.handle, _ptr(a), ext_k, _ptr(_fe(ext_omega_inv)), _ptr(_fe(ext_divisor)), _ptr(_fe(zeta_inv))))
        return a

    def extended_to_coeff_dev(self, dptr: int, ext_k: int, ext_omega_inv: np.ndarray, ext_divisor: np.ndarray, zeta_inv: np.ndarray):
        self._chk(self.lib.h2hip_extended_to_coeff_dev(self.handle, _vp(dptr), ext_k, _ptr(_fe(ext_omega_inv)), _ptr(_fe(ext_divisor)), _ptr(_fe(zeta_inv))))

    # -- K8 witness-column batches (device pointers)
    def _binop(self, fn, a: np.ndarray, b: np.ndarray) -> np.ndarray:
        a, b = _fe(a), _fe(b)
        assert a.shape == b.shape
        da, db = self.to_device(a), self.to_device(b)
        try:
            self._chk(fn(self.handle, _vp(da), _vp(da), _vp(db), len(a)))
            return self.download(da, a.shape)
        finally:
            self.free(da)
            self.free(db)

    def fr_add(self, a, b):
        return self._binop(self.lib.h2hip_fr_add_batch_dev, a, b)

    def fr_sub(self, a, b):
        return self._binop(self.lib.h2hip_fr_sub_batch_dev, a, b)

    def fr_mul(self, a, b):
        return self._binop(self.lib.h2hip_fr_mul_batch_dev, a, b)

    def fr_mul_add(self, a, b, c):
        a, b, c = _fe(a), _fe(b), _fe(c)
        da, db, dc = self.to_device(a), self.to_device(b), self.to_device(c)
        try:
            self._chk(self.lib.h2hip_fr_mul_add_batch_dev(self.handle, _vp(da), _vp(da), _vp(db), _vp(dc), len(a)))
            return self.download(da, a.shape)
        finally:
            for d in (da, db, dc):
                self.free(d)

    def bench_modmul(self, blocks: int = 4096, iters: int = 512, chains: int = 1, unsaturated: bool = False):
        """returns (elapsed_ms, modmuls) of the multiplier probe kernel (saturated 8x32 or unsaturated 9x29 limbs)"""
        ms, mm = C.c_double(), C.c_double()
        fn = self.lib.h2hip_bench_modmul29 if unsaturated else self.lib.h2hip_bench_modmul
        self._chk(fn(self.handle, blocks, iters, chains, C.byref(ms), C.byref(mm)))
        return ms.value, mm.value

    def bench_gather(self, kind: int, table_bytes: int, lanes: int, per_lane: int):
        """(elapsed_ms, useful_bytes) of the HBM calibration probe: kind 0 = stream, 64 / 128 = random gathers of aligned entries"""
        ms, nb = C.c_double(), C.c_double()
        self._chk(self.lib.h2hip_bench_gather(self.handle, kind, table_bytes, lanes, per_lane, C.byref(ms), C.byref(nb)))
        return ms.value, nb.value

    # -- K4/K5/K7 (host-array conveniences over the _dev entry points)
    def fr_axpy(self, y: np.ndarray, a: np.ndarray, x: np.ndarray) -> np.ndarray:
        """y + a*x (a: one field element)"""
        y, x = _fe(y), _fe(x)
        assert y.shape == x.shape
        dy, dx = self.to_device(y), self.to_device(x)
        try:
            self._chk(self.lib.h2hip_fr_axpy_dev(self.handle, _vp(dy), _ptr(_fe(a)), _vp(dx), len(y)))
            return self.download(dy, y.shape)
        finally:
            self.free(dy)
            self.free(dx)

    def fr_scale(self, y: np.ndarray, s: np.ndarray) -> np.ndarray:
        y = _fe(y)
        dy = self.to_device(y)
        try:
            self._chk(self.lib.h2hip_fr_scale_dev(self.handle, _vp(dy), _ptr(_fe(s)), len(y)))
            return self.download(dy, y.shape)
        finally:
            self.free(dy)

    def fr_batch_invert(self, a: np.ndarray) -> np.ndarray:
        a = _fe(a)
        d = self.to_device(a)
        try:
            self._chk(self.lib.h2hip_fr_batch_invert_dev(self.handle, _vp(d), len(a)))
            return self.download(d, a.shape)
        finally:
            self.free(d)

    def fr_prefix_product(self, a: np.ndarray) -> np.ndarray:
        a = _fe(a)
        d, o = self.to_device(a), self.malloc(max(a.nbytes, 32))
        try:
            self._chk(self.lib.h2hip_fr_prefix_product_dev(self.handle, _vp(o), _vp(d), len(a)))
            return self.download(o, a.shape)
        finally:
            self.free(d)
            self.free(o)

    def fr_grand_product(self, num: np.ndarray, den: np.ndarray) -> np.ndarray:
        num, den = _fe(num), _fe(den)
        dn, dd, z = self.to_device(num), self.to_device(den), self.malloc(32 * (len(num) + 1))
        try:
            self._chk(self.lib.h2hip_fr_grand_product_dev(self.handle, _vp(z), _vp(dn), _vp(dd), len(num)))
            return self.download(z, (len(num) + 1, 4))
        finally:
            for d in (dn, dd, z):
                self.free(d)

    def fr_grand_products(self, nums, dens, chained: bool):
        """several grand products of equal length at once -> list of (len + 1, 4) arrays; chained: product s starts at the last value of s - 1"""
        nums, dens = [_fe(a) for a in nums], [_fe(a) for a in dens]
        seg = len(nums[0]) if nums else 0
        if any(len(a) != seg for a in nums + dens) or len(nums) != len(dens):
            raise ValueError("segments of equal length")
        cat = lambda cols: np.concatenate(cols) if cols and seg else np.zeros((1, 4), dtype=np.uint64)
        dn, dd = self.to_device(cat(nums)), self.to_device(cat(dens))
        zs = [self.malloc(32 * (seg + 1)) for _ in nums]
        try:
            arr = (_vp * max(len(zs), 1))(*[_vp(z) for z in zs])
            self._chk(self.lib.h2hip_fr_grand_products_dev(self.handle, arr, _vp(dn), _vp(dd), len(zs), seg, 1 if chained else 0))
            return [self.download(z, (seg + 1, 4)) for z in zs]
        finally:
            for d in [dn, dd] + zs:
                self.free(d)

    def fr_eval_polynomial(self, coeffs: np.ndarray, x: np.ndarray) -> np.ndarray:
        c = _fe(coeffs)
        d = self.to_device(c) if len(c) else self.malloc(32)
        out = np.zeros((1, 4), dtype=np.uint64)
        try:
            self._chk(self.lib.h2hip_fr_eval_polynomial_dev(self.handle, _vp(d), len(c), _ptr(_fe(x)), _ptr(out)))
            return out
        finally:
            self.free(d)

    def fr_kate_division(self, coeffs: np.ndarray, b: np.ndarray) -> np.ndarray:
        c = _fe(coeffs)
        d, q = self.to_device(c), self.malloc(32 * max(len(c) - 1, 1))
        try:
            self._chk(self.lib.h2hip_fr_kate_division_dev(self.handle, _vp(q), _vp(d), len(c), _ptr(_fe(b))))
            return self.download(q, (len(c) - 1, 4))
        finally:
            self.free(d)
            self.free(q)

    # -- prover steps between the big kernels (host-array conveniences for tests)
    def assigned_resolve(self, num: np.ndarray, den: np.ndarray) -> np.ndarray:
        """Assigned::Rational columns -> values: num * den^-1, 0^-1 := 0"""
        num, den = _fe(num), _fe(den)
        assert num.shape == den.shape
        dn, dd, do = self.to_device(num), self.to_device(den), self.malloc(max(num.nbytes, 32))
        try:
            self._chk(self.lib.h2hip_assigned_resolve_dev(self.handle, _vp(do), _vp(dn), _vp(dd), len(num)))
            return self.download(do, num.shape)
        finally:
            for d in (dn, dd, do):
                self.free(d)

    def fr_axpby(self, y: np.ndarray, s: np.ndarray, a: np.ndarray, x: np.ndarray) -> np.ndarray:
        """s*y + a*x"""
        y, x = _fe(y), _fe(x)
        dy, dx = self.to_device(y), self.to_device(x)
        try:
            self._chk(self.lib.h2hip_fr_axpby_dev(self.handle, _vp(dy), _ptr(_fe(s)), _ptr(_fe(a)), _vp(dx), len(y)))
            return self.download(dy, y.shape)
        finally:
            self.free(dy)
            self.free(dx)

    def fr_linear_combination(self, polys, coeffs: np.ndarray) -> np.ndarray:
        """sum_j coeffs[j] * polys[j]  (columns of equal length)"""
        cols = [_fe(p) for p in polys]
        coeffs = _fe(coeffs)
        if len(coeffs) != len(cols) or any(len(c) != len(cols[0]) for c in cols):
            raise ValueError("one coefficient per polynomial, polynomials of equal length")
        n = len(cols[0]) if cols else 0
        dptrs = [self.to_device(c) for c in cols]
        do = self.malloc(max(32 * n, 32))
        try:
            arr = (_vp * max(len(cols), 1))(*[_vp(d) for d in dptrs])
            self._chk(self.lib.h2hip_fr_linear_combination_dev(self.handle, _vp(do), arr, _ptr(coeffs) if len(cols) else None, len(cols), n))
            return self.download(do, (n, 4))
        finally:
            for d in dptrs + [do]:
                self.free(d)

    def fr_sub_low(self, y: np.ndarray, low: np.ndarray) -> np.ndarray:
        y, low = _fe(y), _fe(low)
        dy = self.to_device(y)
        try:
            self._chk(self.lib.h2hip_fr_sub_low_dev(self.handle, _vp(dy), _ptr(low), len(low)))
            return self.download(dy, y.shape)
        finally:
            self.free(dy)

    def fr_eval_polynomial_batch(self, polys, points: np.ndarray) -> np.ndarray:
        """[poly_j(points[j])]: one pass over all pairs"""
        cols = [_fe(p) for p in polys]
        pts = _fe(points)
        assert len(cols) == len(pts)
        d = [self.to_device(c) if len(c) else self.malloc(32) for c in cols]
        out = np.zeros((len(cols), 4), dtype=np.uint64)
        try:
            arr = (_vp * max(len(d), 1))(*[_vp(p) for p in d])
            lens = (_sz * max(len(d), 1))(*[len(c) for c in cols])
            self._chk(self.lib.h2hip_fr_eval_polynomial_batch_dev(self.handle, arr, lens, _ptr(pts), len(cols), _ptr(out)))
            return out
        finally:
            for p in d:
                self.free(p)

    def permutation_product_terms(self, cols, sigmas, first_col_index: int, beta, gamma, delta, omega):
        """(num, den) of one permutation set over all rows of the given columns"""
        cols, sigmas = [_fe(c) for c in cols], [_fe(c) for c in sigmas]
        rows, m = len(cols[0]), len(cols)
        dc, ds = [self.to_device(c) for c in cols], [self.to_device(c) for c in sigmas]
        dn, dd = self.malloc(32 * max(rows, 1)), self.malloc(32 * max(rows, 1))
        try:
            pc, ps = (_vp * m)(*[_vp(p) for p in dc]), (_vp * m)(*[_vp(p) for p in ds])
            self._chk(self.lib.h2hip_permutation_product_terms_dev(self.handle, _vp(dn), _vp(dd), pc, ps, m, first_col_index, rows, _ptr(_fe(beta)),
                                                                   _ptr(_fe(gamma)), _ptr(_fe(delta)), _ptr(_fe(omega))))
            return self.download(dn, (rows, 4)), self.download(dd, (rows, 4))
        finally:
            for p in dc + ds + [dn, dd]:
                self.free(p)

    def lookup_product_terms(self, a, s, a_perm, s_perm, beta, gamma):
        arrs = [_fe(v) for v in (a, s, a_perm, s_perm)]
        rows = len(arrs[0])
        d = [self.to_device(v) for v in arrs]
        dn, dd = self.malloc(32 * max(rows, 1)), self.malloc(32 * max(rows, 1))
        try:
            self._chk(self.lib.h2hip_lookup_product_terms_dev(self.handle, _vp(dn), _vp(dd), *[_vp(p) for p in d], rows, _ptr(_fe(beta)), _ptr(_fe(gamma))))
            return self.download(dn, (rows, 4)), self.download(dd, (rows, 4))
        finally:
            for p in d + [dn, dd]:
                self.free(p)

    def fr_kate_division_multi_acc(self, acc: np.ndarray, coeffs: np.ndarray, points: np.ndarray, weights: np.ndarray) -> np.ndarray:
        """acc[0..n-1) + sum_j weights[j] * (f(X) - f(points[j])) / (X - points[j])"""
        c, pts, w, a = _fe(coeffs), _fe(points), _fe(weights), _fe(acc)
        d, q = self.to_device(c), self.to_device(a)
        try:
            self._chk(self.lib.h2hip_fr_kate_division_multi_acc_dev(self.handle, _vp(q), _vp(d), len(c), _ptr(pts), _ptr(w), len(pts)))
            return self.download(q, (len(a), 4))
        finally:
            self.free(d)
            self.free(q)

    def fr_kate_division_sets(self, polys, point_sets, weight_sets) -> np.ndarray:
        """sum over the sets i of sum_j weight_sets[i][j] * (polys[i](X) - polys[i](p_ij)) / (X - p_ij): one call for all sets"""
        n = len(_fe(polys[0]))
        dp, pp = self._ptr_table(polys)
        pts = np.concatenate([_fe(p) for p in point_sets])
        ws = np.concatenate([_fe(w) for w in weight_sets])
        sizes = (_u32 * len(polys))(*[len(_fe(p)) for p in point_sets])
        q = self.malloc(32 * max(n - 1, 1))
        try:
            self._chk(self.lib.h2hip_fr_kate_division_sets_dev(self.handle, _vp(q), pp, n, _ptr(pts), _ptr(ws), sizes, len(polys), 0))
            return self.download(q, (n - 1, 4))
        finally:
            for d in dp:
                self.free(d)
            self.free(q)

    def fr_kate_division_range(self, coeffs: np.ndarray, points: np.ndarray, weights: np.ndarray, carries: np.ndarray) -> np.ndarray:
        """the quotient coefficients lo .. lo + n - 1 of sum_j weights[j] * f(X) / (X - points[j]) from f's coefficient range [lo, lo + n) and the
        carries of the ranges above (h2hip.h)"""
        c, pts, w, cr = _fe(coeffs), _fe(points), _fe(weights), _fe(carries)
        d, q = self.to_device(c), self.malloc(32 * len(c))
        try:
            self._chk(self.lib.h2hip_fr_kate_division_range_dev(self.handle, _vp(q), _vp(d), len(c), _ptr(pts), _ptr(w), _ptr(cr), len(pts)))
            return self.download(q, (len(c), 4))
        finally:
            self.free(d)
            self.free(q)

    def fr_kate_division_multi(self, coeffs: np.ndarray, points: np.ndarray, weights: np.ndarray) -> np.ndarray:
        """sum_j weights[j] * (f(X) - f(points[j])) / (X - points[j])"""
        c, pts, w = _fe(coeffs), _fe(points), _fe(weights)
        d, q = self.to_device(c), self.malloc(32 * max(len(c) - 1, 1))
        try:
            self._chk(self.lib.h2hip_fr_kate_division_multi_dev(self.handle, _vp(q), _vp(d), len(c), _ptr(pts), _ptr(w), len(pts)))
            return self.download(q, (len(c) - 1, 4))
        finally:
            self.free(d)
            self.free(q)

    def quotient_flex_gate(self, acc: np.ndarray, q: np.ndarray, a: np.ndarray, ext_k: int, k: int, y: np.ndarray) -> np.ndarray:
        acc, q, a = _fe(acc), _fe(q), _fe(a)
        da, dq, dv = self.to_device(acc), self.to_device(q), self.to_device(a)
        try:
            self._chk(self.lib.h2hip_quotient_flex_gate_dev(self.handle, _vp(da), _vp(dq), _vp(dv), ext_k, k, _ptr(_fe(y))))
            return self.download(da, acc.shape)
        finally:
            for d in (da, dq, dv):
                self.free(d)

    def divide_by_vanishing_poly(self, a: np.ndarray, ext_k: int, k: int, ext_omega: np.ndarray, zeta: np.ndarray) -> np.ndarray:
        a = _fe(a)
        da = self.to_device(a)
        try:
            self._chk(self.lib.h2hip_divide_by_vanishing_poly_dev(self.handle, _vp(da), ext_k, k, _ptr(_fe(ext_omega)), _ptr(_fe(zeta))))
            return self.download(da, a.shape)
        finally:
            self.free(da)

    # -- K8 Poseidon
    def poseidon_set_spec(self, t: int, r_f: int, r_p: int, round_constants: np.ndarray, mds: np.ndarray):
        rc, m = _fe(round_constants), _fe(mds)
        assert len(rc) == (r_f + r_p) * t and len(m) == t * t
        self._chk(self.lib.h2hip_poseidon_set_spec(self.handle, t, r_f, r_p, _ptr(rc), _ptr(m)))
        self._pos_t = t

    def poseidon_permute(self, states: np.ndarray, inputs: Optional[np.ndarray] = None) -> np.ndarray:
        """states: (n, t, 4); inputs: (n, m, 4) with m <= t-1 or None"""
        t = self._pos_t
        st = np.ascontiguousarray(states, dtype=np.uint64).reshape(-1, t, 4)
        n = len(st)
        m = 0 if inputs is None else np.asarray(inputs).reshape(n, -1, 4).shape[1]
        ds = self.to_device(st)
        di = self.to_device(np.ascontiguousarray(inputs, dtype=np.uint64)) if m else None
        try:
            self._chk(self.lib.h2hip_poseidon_permute_batch_dev(self.handle, _vp(ds), _vp(di) if di else None, m, n))
            return self.download(ds, st.shape)
        finally:
            self.free(ds)
            if di:
                self.free(di)

    # -- K6 lookup / permutation identities (host-array conveniences for tests)
    def quotient_lookup(self, acc, z, a, s, a_perm, s_perm, l0, l_last, l_blind, ext_k, k, beta, gamma, y) -> np.ndarray:
        arrs = [_fe(v) for v in (acc, z, a, s, a_perm, s_perm, l0, l_last, l_blind)]
        d = [self.to_device(v) for v in arrs]
        try:
            self._chk(self.lib.h2hip_quotient_lookup_dev(self.handle, *[_vp(p) for p in d], ext_k, k, _ptr(_fe(beta)), _ptr(_fe(gamma)), _ptr(_fe(y))))
            return self.download(d[0], arrs[0].shape)
        finally:
            for p in d:
                self.free(p)

    def quotient_permutation_set(self, acc, z, z_prev, cols, sigmas, first_col_index, l0, l_last, l_blind, ext_k, k, terms,
                                 last_rotation, beta, gamma, delta, zeta, ext_omega, y) -> np.ndarray:
        d_acc, d_z = self.to_device(_fe(acc)), self.to_device(_fe(z))
        d_zp = self.to_device(_fe(z_prev)) if z_prev is not None else None
        d_cols = [self.to_device(_fe(c)) for c in cols]
        d_sig = [self.to_device(_fe(c)) for c in sigmas]
        d_l = [self.to_device(_fe(v)) for v in (l0, l_last, l_blind)]
        n = len(cols)
        pc, ps = (_vp * n)(*[_vp(p) for p in d_cols]), (_vp * n)(*[_vp(p) for p in d_sig])
        try:
            self._chk(self.lib.h2hip_quotient_permutation_set_dev(
                self.handle, _vp(d_acc), _vp(d_z), _vp(d_zp) if d_zp else None, pc, ps, n, first_col_index, _vp(d_l[0]), _vp(d_l[1]), _vp(d_l[2]),
                ext_k, k, int(terms), int(last_rotation), _ptr(_fe(beta)), _ptr(_fe(gamma)), _ptr(_fe(delta)), _ptr(_fe(zeta)),
                _ptr(_fe(ext_omega)), _ptr(_fe(y))))
            return self.download(d_acc, _fe(acc).shape)
        finally:
            for p in [d_acc, d_z] + ([d_zp] if d_zp else []) + d_cols + d_sig + d_l:
                self.free(p)

    def permutation_product_terms_sets(self, cols, sigmas, chunk_len, beta, gamma, delta, omega, row0=None, rows=None):
        """(num, den) factor columns of every permutation set: two lists of (rows, 4) arrays; row0 / rows: only that row range
        (h2hip_permutation_product_terms_rows_dev)"""
        by_rows = row0 is not None
        row0 = row0 or 0
        rows = len(_fe(cols[0])) - row0 if rows is None else rows
        sets = (len(cols) + chunk_len - 1) // chunk_len
        dc, pc = self._ptr_table(cols)
        ds, ps = self._ptr_table(sigmas)
        dn, dd = self.malloc(32 * max(rows * sets, 1)), self.malloc(32 * max(rows * sets, 1))
        try:
            if by_rows:
                self._chk(self.lib.h2hip_permutation_product_terms_rows_dev(self.handle, _vp(dn), _vp(dd), pc, ps, len(dc), chunk_len, row0, rows,
                                                                            _ptr(_fe(beta)), _ptr(_fe(gamma)), _ptr(_fe(delta)), _ptr(_fe(omega))))
            else:
                self._chk(self.lib.h2hip_permutation_product_terms_sets_dev(self.handle, _vp(dn), _vp(dd), pc, ps, len(dc), chunk_len, rows, _ptr(_fe(beta)),
                                                                            _ptr(_fe(gamma)), _ptr(_fe(delta)), _ptr(_fe(omega))))
            num, den = self.download(dn, (sets * rows, 4)), self.download(dd, (sets * rows, 4))
            return [num[s * rows:(s + 1) * rows] for s in range(sets)], [den[s * rows:(s + 1) * rows] for s in range(sets)]
        finally:
            for p in dc + ds + [dn, dd]:
                self.free(p)

    def lookup_permute_batch(self, inputs, table: np.ndarray, usable_rows: int):
        """[(a_perm, s_perm)] for several input columns against one table (sorted once): h2hip_lookup_permute_presorted_batch_dev"""
        table = _fe(table)
        da, pa = self._ptr_table(inputs)
        dt = self.to_device(table)
        dsorted = self.malloc(max(self.lib.h2hip_lookup_sorted_table_bytes(usable_rows), 32))
        outs = [self.malloc(max(table.nbytes, 32)) for _ in range(2 * len(da))]
        try:
            if usable_rows:
                self._chk(self.lib.h2hip_lookup_table_sort_dev(self.handle, _vp(dt), usable_rows, _vp(dsorted)))
            pap = (_vp * max(len(da), 1))(*[_vp(p) for p in outs[0::2]])
            psp = (_vp * max(len(da), 1))(*[_vp(p) for p in outs[1::2]])
            self._chk(self.lib.h2hip_lookup_permute_presorted_batch_dev(self.handle, pa, _vp(dsorted), usable_rows, pap, psp, len(da)))
            return [(self.download(outs[2 * j], table.shape)[:usable_rows], self.download(outs[2 * j + 1], table.shape)[:usable_rows])
                    for j in range(len(da))]
        finally:
            for p in da + [dt, dsorted] + outs:
                self.free(p)

    def _ptr_table(self, cols):
        d = [self.to_device(_fe(c)) for c in cols]
        return d, (_vp * max(len(d), 1))(*[_vp(p) for p in d])

    def quotient_flex_gate_batch(self, acc, qs, advs, ext_k, k, y) -> np.ndarray:
        """acc folded with q_j*(a + b*c - d) for every (q_j, a_j) in order: one launch per 64 columns"""
        d_acc = self.to_device(_fe(acc))
        dq, pq = self._ptr_table(qs)
        da, pa = self._ptr_table(advs)
        try:
            self._chk(self.lib.h2hip_quotient_flex_gate_batch_dev(self.handle, _vp(d_acc), pq, pa, len(dq), ext_k, k, _ptr(_fe(y))))
            return self.download(d_acc, _fe(acc).shape)
        finally:
            for p in [d_acc] + dq + da:
                self.free(p)

    def quotient_lookups(self, acc, zs, a_s, s_s, aps, sps, l0, l_last, l_blind, ext_k, k, beta, gamma, y) -> np.ndarray:
        """acc folded with the five identities of every lookup in order: one launch per 32 lookups"""
        d_acc = self.to_device(_fe(acc))
        tabs = [self._ptr_table(c) for c in (zs, a_s, s_s, aps, sps)]
        d_l = [self.to_device(_fe(v)) for v in (l0, l_last, l_blind)]
        try:
            self._chk(self.lib.h2hip_quotient_lookups_dev(self.handle, _vp(d_acc), *[t[1] for t in tabs], len(zs), _vp(d_l[0]), _vp(d_l[1]), _vp(d_l[2]),
                                                          ext_k, k, _ptr(_fe(beta)), _ptr(_fe(gamma)), _ptr(_fe(y))))
            return self.download(d_acc, _fe(acc).shape)
        finally:
            for p in [d_acc] + d_l + [q for t in tabs for q in t[0]]:
                self.free(p)

    def quotient_permutation_sets(self, acc, zs, cols, sigmas, chunk_len, l0, l_last, l_blind, ext_k, k, last_rotation, beta, gamma, delta, zeta,
                                  ext_omega, y) -> np.ndarray:
        """the whole permutation argument (all sets, evaluate_h's order) folded into acc"""
        d_acc = self.to_device(_fe(acc))
        dz, pz = self._ptr_table(zs)
        dc, pc = self._ptr_table(cols)
        ds, ps = self._ptr_table(sigmas)
        d_l = [self.to_device(_fe(v)) for v in (l0, l_last, l_blind)]
        try:
            self._chk(self.lib.h2hip_quotient_permutation_sets_dev(
                self.handle, _vp(d_acc), pz, len(dz), pc, ps, len(dc), chunk_len, _vp(d_l[0]), _vp(d_l[1]), _vp(d_l[2]), ext_k, k, int(last_rotation),
                _ptr(_fe(beta)), _ptr(_fe(gamma)), _ptr(_fe(delta)), _ptr(_fe(zeta)), _ptr(_fe(ext_omega)), _ptr(_fe(y))))
            return self.download(d_acc, _fe(acc).shape)
        finally:
            for p in [d_acc] + dz + dc + ds + d_l:
                self.free(p)

    def lookup_permute(self, a: np.ndarray, s: np.ndarray, usable_rows: int, presort_table: bool = False):
        """(a_perm, s_perm) over rows [0, usable_rows); presort_table: go through h2hip_lookup_table_sort_dev + _permute_presorted_dev (what a
        prover with a cached, sorted table column does)"""
        a, s = _fe(a), _fe(s)
        if not (len(a) == len(s) and 0 <= usable_rows <= len(a)):
            raise ValueError("lookup_permute: need len(a) == len(s) >= usable_rows")
        da, ds = self.to_device(a), self.to_device(s)
        dap, dsp = self.malloc(max(a.nbytes, 32)), self.malloc(max(a.nbytes, 32))
        dsorted = None
        try:
            if presort_table and usable_rows:
                dsorted = self.malloc(self.lib.h2hip_lookup_sorted_table_bytes(usable_rows))
                self._chk(self.lib.h2hip_lookup_table_sort_dev(self.handle, _vp(ds), usable_rows, _vp(dsorted)))
                self._chk(self.lib.h2hip_lookup_permute_presorted_dev(self.handle, _vp(da), _vp(dsorted), usable_rows, _vp(dap), _vp(dsp)))
            else:
                self._chk(self.lib.h2hip_lookup_permute_dev(self.handle, _vp(da), _vp(ds), usable_rows, _vp(dap), _vp(dsp)))
            return self.download(dap, a.shape)[:usable_rows], self.download(dsp, a.shape)[:usable_rows]
        finally:
            for d in (da, ds, dap, dsp) + ((dsorted,) if dsorted else ()):
                self.free(d)
