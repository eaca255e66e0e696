"""ctypes binding of include/fdsr.h.  Fails loudly when libfdsr_hip.so is missing:
there is no CPU or PyTorch fallback behind this package."""
import ctypes as C
import os

from .build import LIB

FDSR_MAX_MULTS = 8
FDSR_SAMPLE_GRAPH = 1
FDSR_METRIC_FIELDS = 8
FDSR_SSIM_UNIFORM7, FDSR_SSIM_GAUSS11 = 1, 2
PRECISIONS = {'f32': 0, 'f16x3': 1, 'bf16': 2, 'f16': 3}


class FdsrConfig(C.Structure):
    _fields_ = [('in_channel', C.c_int32), ('out_channel', C.c_int32), ('inner_channel', C.c_int32),
                ('norm_groups', C.c_int32), ('n_mults', C.c_int32), ('channel_mults', C.c_int32 * FDSR_MAX_MULTS),
                ('res_blocks', C.c_int32), ('dropout', C.c_float), ('image_size', C.c_int32), ('variant', C.c_int32),
                ('n_attn_res', C.c_int32), ('attn_res', C.c_int32 * FDSR_MAX_MULTS)]


class FdsrSchedule(C.Structure):
    _fields_ = [('n_timestep', C.c_int32), ('noise_level', C.POINTER(C.c_float)), ('sqrt_recip', C.POINTER(C.c_float)),
                ('sqrt_recipm1', C.POINTER(C.c_float)), ('coef1', C.POINTER(C.c_float)), ('coef2', C.POINTER(C.c_float)),
                ('sigma', C.POINTER(C.c_float))]


class FdsrError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f'libfdsr_hip error {code}: {msg}')
        self.code = code


class FdsrSaturated(FdsrError):
    """f16x3: a raw convolution input left the f16 range (FDSR_E_SATURATED): the call's output is not fp32-grade."""


FDSR_E_SATURATED = -6

# include/fdsr.h: enum fdsr_k32_bits / fdsr_strip_bits (tests/test_abi_symbols.py checks the two against the header)
K32 = {'F16X3': 1, 'BF16': 2, 'RIDER_16ROW': 4, 'SMALL_GRID_2ROW': 8, 'UP2': 16, 'SMALL_WG_F16X3': 32, 'SMALL_WG_RIDER_F16X3': 64,
       'SMALL_WG_BF16': 128, 'SMALL_WG_RIDER_BF16': 512, 'RIDER_FIRST_8WAVE': 1024}
K32_DEFAULT = 1 | 2 | 8 | 16 | 32 | 64 | 128 | 1024
STRIP = {'BF16_64': 1, 'F16X3_64': 2, 'BF16_ONE_WG': 4, 'BF16_CAT64': 8, 'BF16_RIDER': 16, 'BF16_CAT128': 32, 'BF16_COUT128': 64}
STRIP_DEFAULT = 1 | 2 | 8 | 16 | 64
STRIP_ALL = 127


def bit_names(table, value):
    """'F16X3|BF16|...' for a k32 / strip option value."""
    return '|'.join(n for n, b in table.items() if value & b) or '0'


_lib = None

# name -> (restype, argtypes); every symbol include/fdsr.h declares
SYMBOLS = {
    'fdsr_create': (C.c_int, [C.POINTER(FdsrConfig), C.POINTER(C.c_void_p)]),
    'fdsr_destroy': (None, [C.c_void_p]),
    'fdsr_last_error': (C.c_char_p, [C.c_void_p]),
    'fdsr_version': (C.c_char_p, []),
    'fdsr_num_weights': (C.c_int, [C.c_void_p]),
    'fdsr_weight_info': (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int),
                                   C.POINTER(C.c_int)]),
    'fdsr_load_weight': (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int]),
    'fdsr_weights_complete': (C.c_int, [C.c_void_p]),
    'fdsr_set_schedule': (C.c_int, [C.c_void_p, C.POINTER(FdsrSchedule)]),
    'fdsr_workspace_bytes': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    'fdsr_unet_forward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                    C.c_size_t, C.c_void_p]),
    'fdsr_sample': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                              C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]),
    'fdsr_set_precision': (C.c_int, [C.c_void_p, C.c_int]),
    'fdsr_set_seed': (C.c_int, [C.c_void_p, C.c_uint64]),
    'fdsr_debug_tensor_elem_bytes': (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p]),
    'fdsr_randn': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'fdsr_resize_bicubic_u8': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p]),
    'fdsr_tensor2img_u8': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                     C.c_void_p]),
    'fdsr_u8_to_tensor': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                    C.c_void_p]),
    'fdsr_image_metrics_workspace_bytes': (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    'fdsr_image_metrics_u8': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                        C.c_void_p, C.c_size_t, C.c_void_p]),
    'fdsr_set_debug': (C.c_int, [C.c_void_p, C.c_int]),
    'fdsr_debug_option': (C.c_int, [C.c_char_p, C.c_longlong]),
    'fdsr_check_saturation': (C.c_int, [C.c_void_p, C.c_void_p]),
    'fdsr_debug_tensor': (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                    C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    'fdsr_set_training': (C.c_int, [C.c_void_p, C.c_int]),
    'fdsr_set_dropout_seed': (C.c_int, [C.c_void_p, C.c_uint64]),
    'fdsr_debug_dropout_mask': (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                          C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_float)]),
    'fdsr_train_workspace_bytes': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    'fdsr_train_grads': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.POINTER(C.c_float), C.c_int,
                                   C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    'fdsr_train_grads_pairs': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.POINTER(C.c_float),
                                         C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    'fdsr_adam_step': (C.c_int, [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    'fdsr_get_weight': (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p]),
    'fdsr_get_grad': (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p]),
    'fdsr_sync_weight_forms': (C.c_int, [C.c_void_p]),
    'fdsr_get_optimizer_state': (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]),
    'fdsr_set_optimizer_state': (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_int]),
    'fdsr_grad_arena': (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    'fdsr_profile_begin': (C.c_int, [C.c_void_p]),
    'fdsr_profile_end': (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                   C.POINTER(C.c_double)]),
}


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(os.environ.get('FDSR_LIB', LIB)):
        raise ImportError(
            f'{LIB} is missing: build it with `python -m fastdiffsr_amd.build` (or __graft_entry__.build()). '
            'fastdiffsr_amd has no CPU/PyTorch fallback for the sampling path.')
    # torch first: it ships its own libamdhip64; loading ours before it would bind this library to the
    # system copy and leave two HIP runtimes in the process (ours would then see no device)
    import torch  # noqa: F401
    lib = C.CDLL(os.environ.get('FDSR_LIB', LIB))   # FDSR_LIB: A/B a differently built engine on one box
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)      # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def debug_option(name, value):
    """Launcher A/B options (include/fdsr.h: fdsr_debug_option); process-wide."""
    rc = load().fdsr_debug_option(name.encode(), int(value))
    if rc != 0:
        raise FdsrError(rc, f'unknown debug option {name!r}')


def check(handle, rc):
    if rc != 0:
        msg = load().fdsr_last_error(handle)
        raise FdsrError(rc, msg.decode() if msg else '?')
