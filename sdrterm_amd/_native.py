"""ctypes binding of libsdrterm_hip.so -- the only way this package computes anything.

Loading fails loudly (``NativeLibraryError``) when the shared library has not been built; there is no
fallback.  Build it with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C sdrterm_amd/csrc``.
"""
import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
# SDRTERM_HIP_LIB: another build of the same library (A/B measurements of kernel variants); default: the in-tree build
LIB_PATH = os.environ.get('SDRTERM_HIP_LIB') or os.path.join(_HERE, 'libsdrterm_hip.so')

# enum sdrt_dtype, keyed by the reference's DataType names (src/misc/file_util.py:44-56)
DTYPES = {'b': 0, 'B': 1, 'h': 2, 'H': 3, 'i': 4, 'I': 5, 'f': 6, 'd': 7}
ITEM_SIZE = {'b': 1, 'B': 1, 'h': 2, 'H': 2, 'i': 4, 'I': 4, 'f': 4, 'd': 8}
MODES = {'fm': 0, 'am': 1, 're': 2, 'im': 3}
STAGE_A = {'auto': 0, 'i8': 1, 'f64': 2}      # enum sdrt_stage_a
OK, EINVAL, EHIP, ENOMEM, ENODEVICE = 0, 1, 2, 3, 4

ABI_VERSION = 7
EXPORTS = (
    'sdrt_abi_version', 'sdrt_config_size', 'sdrt_device_count', 'sdrt_last_error', 'sdrt_create', 'sdrt_destroy',
    'sdrt_out_per_chunk', 'sdrt_rows', 'sdrt_process_device', 'sdrt_process_host',
    'sdrt_process_device_debug', 'sdrt_get_iq_offset', 'sdrt_set_iq_offset', 'sdrt_profile_enable',
    'sdrt_kernel_time', 'sdrt_profile_reset', 'sdrt_kernel_name', 'sdrt_malloc', 'sdrt_free',
    'sdrt_memcpy_h2d', 'sdrt_memcpy_d2h', 'sdrt_memcpy_d2d', 'sdrt_device_synchronize', 'sdrt_set_device',
    'sdrt_selftest_mfma', 'sdrt_selftest_mfma_i8', 'sdrt_set_smoothing', 'sdrt_stage_a_kind', 'sdrt_debug_stamps', 'sdrt_demodulate_host',
    'sdrt_stream_create', 'sdrt_stream_destroy', 'sdrt_stream_synchronize', 'sdrt_host_alloc', 'sdrt_host_free',
    'sdrt_memcpy_h2d_async', 'sdrt_memcpy_d2h_async', 'sdrt_stream_wait_stream',
    'sdrt_comm_unique_id', 'sdrt_comm_init', 'sdrt_comm_bcast', 'sdrt_comm_bcast_scatter', 'sdrt_comm_allgather', 'sdrt_comm_world', 'sdrt_comm_rank',
    'sdrt_comm_destroy', 'sdrt_submit_host', 'sdrt_wait', 'sdrt_spectrum_host', 'sdrt_stft_host', 'sdrt_iq_range_device',
    'sdrt_event_create', 'sdrt_event_destroy', 'sdrt_event_record', 'sdrt_stream_wait_event', 'sdrt_event_query',
    'sdrt_event_synchronize', 'sdrt_reserve', 'sdrt_taps_release', 'sdrt_mem_info', 'sdrt_alloc_count',
    'sdrt_event_create_timed', 'sdrt_event_elapsed_ms', 'sdrt_memcpy_d2d_async', 'sdrt_refined_items',
)


class NativeLibraryError(RuntimeError):
    pass


class SdrtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f'[sdrt {code}] {message}')
        self.code = code


class Config(C.Structure):
    """struct sdrt_config (include/sdrterm_hip.h)."""
    _fields_ = [
        ('struct_size', C.c_uint32), ('device', C.c_int32), ('fs', C.c_double),
        ('decimation', C.c_int32), ('mode', C.c_int32), ('dtype', C.c_int32), ('swap_endianness', C.c_int32),
        ('n_samples', C.c_int32), ('n_rows', C.c_int32), ('row_freqs', C.POINTER(C.c_double)),
        ('apply_shift', C.c_int32), ('normalize', C.c_int32), ('norm_xmin', C.c_double),
        ('norm_inv_range', C.c_double), ('correct_iq', C.c_int32), ('reserved0', C.c_int32),
        ('iq_inductance', C.c_double), ('dec_sos', C.POINTER(C.c_double)), ('dec_sections', C.c_int32),
        ('out_sections', C.c_int32), ('out_sos', C.POINTER(C.c_double)), ('row_modes', C.POINTER(C.c_int32)),
        ('stage_a', C.c_int32), ('faithful_f32', C.c_int32), ('f32_sos', C.POINTER(C.c_float)),
        ('f32_zi', C.POINTER(C.c_float)), ('shift_table', C.POINTER(C.c_double)),
    ]


_lib = None


def _share_hip_runtime_with_torch() -> None:
    """One HIP runtime per process.  By default this library binds to the SYSTEM runtime (libamdhip64 as the dynamic linker
    finds it): the product does not depend on any other package's copy.  A PyTorch-ROCm wheel bundles its own libamdhip64 and
    insists on it; a process that wants BOTH this library and torch on the GPU must let torch's runtime come first, so that
    both bind to one copy -- that is either ``import torch`` before this module is used (then the runtime is already loaded
    and this function does nothing), or SDRTERM_TORCH_HIP=1, which pre-loads torch's bundled runtime without importing torch."""
    if 'torch' in sys.modules or os.environ.get('SDRTERM_TORCH_HIP') != '1':
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec('torch')
    except (ImportError, ValueError):
        return
    for root in (spec.submodule_search_locations if spec and spec.submodule_search_locations else ()):
        for name in ('libamdhip64.so', 'libamdhip64.so.7', 'libamdhip64.so.6'):
            path = os.path.join(root, 'lib', name)
            if os.path.exists(path):
                try:
                    C.CDLL(path, mode=C.RTLD_GLOBAL)
                except OSError:
                    pass
                return


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f'{LIB_PATH} is missing: the HIP extension has not been built and there is no CPU path. '
            'Run `make -C sdrterm_amd/csrc` (needs hipcc, targets gfx950).')
    _share_hip_runtime_with_torch()
    try:
        l = C.CDLL(LIB_PATH)
    except OSError as e:  # e.g. libamdhip64 not found
        raise NativeLibraryError(f'cannot load {LIB_PATH}: {e}') from e
    for name in EXPORTS:
        if not hasattr(l, name):
            raise NativeLibraryError(f'{LIB_PATH} does not export {name}')
    l.sdrt_last_error.restype = C.c_char_p
    l.sdrt_kernel_name.restype = C.c_char_p
    l.sdrt_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_void_p)]
    l.sdrt_destroy.argtypes = [C.c_void_p]
    l.sdrt_destroy.restype = None
    l.sdrt_out_per_chunk.argtypes = [C.c_void_p]
    l.sdrt_rows.argtypes = [C.c_void_p]
    l.sdrt_process_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    l.sdrt_process_device_debug.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    l.sdrt_submit_host.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_int64)]
    l.sdrt_wait.argtypes = [C.c_void_p, C.c_int64]
    l.sdrt_spectrum_host.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_double,
                                     C.c_double, C.c_void_p]
    l.sdrt_stft_host.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_double,
                                 C.c_double, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    l.sdrt_process_host.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    l.sdrt_get_iq_offset.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    l.sdrt_set_iq_offset.argtypes = [C.c_void_p, C.c_double, C.c_double]
    l.sdrt_iq_range_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                       C.POINTER(C.c_double), C.c_void_p]
    l.sdrt_profile_enable.argtypes = [C.c_void_p, C.c_int]
    l.sdrt_kernel_time.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    l.sdrt_profile_reset.argtypes = [C.c_void_p]
    l.sdrt_kernel_name.argtypes = [C.c_int]
    l.sdrt_malloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    l.sdrt_free.argtypes = [C.c_void_p]
    l.sdrt_memcpy_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    l.sdrt_memcpy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    l.sdrt_memcpy_d2d.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    l.sdrt_set_device.argtypes = [C.c_int]
    l.sdrt_device_count.argtypes = [C.POINTER(C.c_int)]
    l.sdrt_selftest_mfma.argtypes = [C.POINTER(C.c_double)]
    l.sdrt_selftest_mfma_i8.argtypes = [C.POINTER(C.c_int)]
    l.sdrt_stage_a_kind.argtypes = [C.c_void_p]
    l.sdrt_refined_items.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
    l.sdrt_set_smoothing.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    l.sdrt_demodulate_host.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    l.sdrt_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int]
    l.sdrt_config_size.restype = C.c_size_t
    l.sdrt_stream_create.argtypes = [C.POINTER(C.c_void_p)]
    l.sdrt_stream_destroy.argtypes = [C.c_void_p]
    l.sdrt_stream_synchronize.argtypes = [C.c_void_p]
    l.sdrt_host_alloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    l.sdrt_host_free.argtypes = [C.c_void_p]
    l.sdrt_memcpy_h2d_async.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    l.sdrt_memcpy_d2h_async.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    l.sdrt_stream_wait_stream.argtypes = [C.c_void_p, C.c_void_p]
    l.sdrt_reserve.argtypes = [C.c_void_p, C.c_size_t]
    l.sdrt_event_create.argtypes = [C.POINTER(C.c_void_p)]
    l.sdrt_event_destroy.argtypes = [C.c_void_p]
    l.sdrt_event_record.argtypes = [C.c_void_p, C.c_void_p]
    l.sdrt_stream_wait_event.argtypes = [C.c_void_p, C.c_void_p]
    l.sdrt_event_query.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    l.sdrt_event_synchronize.argtypes = [C.c_void_p]
    l.sdrt_event_create_timed.argtypes = [C.POINTER(C.c_void_p)]
    l.sdrt_event_elapsed_ms.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
    l.sdrt_memcpy_d2d_async.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    l.sdrt_comm_unique_id.argtypes = [C.c_void_p]
    l.sdrt_comm_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    l.sdrt_comm_bcast.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    l.sdrt_comm_bcast_scatter.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    l.sdrt_comm_allgather.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    l.sdrt_comm_world.argtypes = [C.c_void_p]
    l.sdrt_comm_rank.argtypes = [C.c_void_p]
    l.sdrt_comm_destroy.argtypes = [C.c_void_p]
    l.sdrt_comm_destroy.restype = None
    if l.sdrt_abi_version() != ABI_VERSION or l.sdrt_config_size() != C.sizeof(Config):
        raise NativeLibraryError('libsdrterm_hip.so ABI mismatch (version or sdrt_config layout)')
    _lib = l
    return l


def check(rc: int) -> None:
    if rc != OK:
        raise SdrtError(rc, lib().sdrt_last_error().decode(errors='replace'))


def device_count() -> int:
    n = C.c_int(0)
    rc = lib().sdrt_device_count(C.byref(n))
    return n.value if rc == OK else 0


def pinned_array(nbytes: int):
    """A page-locked host buffer (sdrt_host_alloc) as a numpy uint8 array: copies from / to it are asynchronous and run at
    the full PCIe rate.  The memory is released when the array's base object goes away."""
    import numpy as np
    p = C.c_void_p()
    check(lib().sdrt_host_alloc(C.byref(p), int(nbytes)))

    class _Owner:
        def __init__(self, ptr):
            self.ptr = ptr

        def __del__(self):
            try:
                lib().sdrt_host_free(self.ptr)
            except Exception:
                pass
    owner = _Owner(p.value)
    buf = (C.c_uint8 * int(nbytes)).from_address(p.value)
    buf._owner = owner
    return np.frombuffer(buf, dtype=np.uint8)


class DeviceBuffer:
    """A raw device allocation (sdrt_malloc / sdrt_free) for hosts that have no other GPU runtime binding."""

    def __init__(self, nbytes: int):
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        check(lib().sdrt_malloc(C.byref(p), self.nbytes))
        self.ptr = p.value

    def upload(self, data) -> 'DeviceBuffer':
        import numpy as np
        a = np.ascontiguousarray(data)
        assert a.nbytes <= self.nbytes
        check(lib().sdrt_memcpy_h2d(self.ptr, a.ctypes.data, a.nbytes))
        return self

    def download(self, out):
        assert out.flags['C_CONTIGUOUS'] and out.nbytes <= self.nbytes
        check(lib().sdrt_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes))
        return out

    def upload_at(self, offset: int, data) -> 'DeviceBuffer':
        import numpy as np
        a = np.ascontiguousarray(data)
        assert offset + a.nbytes <= self.nbytes
        check(lib().sdrt_memcpy_h2d(self.ptr + int(offset), a.ctypes.data, a.nbytes))
        return self

    def download_at(self, offset: int, out):
        assert out.flags['C_CONTIGUOUS'] and offset + out.nbytes <= self.nbytes
        check(lib().sdrt_memcpy_d2h(out.ctypes.data, self.ptr + int(offset), out.nbytes))
        return out

    def free(self):
        if self.ptr:
            lib().sdrt_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
