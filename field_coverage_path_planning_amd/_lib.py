"""ctypes binding of libfcpp.so (the C ABI declared in include/fcpp.h).

There is no Python or CPU implementation behind these calls: if the shared library is missing, or no
HIP device is usable, the operators raise.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('FCPP_LIBRARY') or os.path.join(_HERE, 'libfcpp.so')   # FCPP_LIBRARY: diagnostic builds only

OK, EINVAL, EHEADLAND, EUNSUPPORTED, EHIP, ENOMEM, ESIZE = 0, -1, -2, -3, -4, -5, -6
TURN_ARC, TURN_CLOTHOID = 0, 1
KIND_SWATH, KIND_UTURN, KIND_HEAD_START, KIND_HEAD_STRAIGHT, KIND_CORNER, KIND_REVERSE, KIND_DETOUR = range(7)
OBSTACLES_FLAG, OBSTACLES_AVOID = 0, 1
RING_AS_VERTICES, RING_REVERSED = 0, 1
KIND_MASK, FLAG_HEADLAND, FLAG_ALAT, FLAG_OUTSIDE, FLAG_OBSTACLE, INDEX_SHIFT = 7, 8, 16, 32, 64, 8
OUTPUT_PITCH = 24 << 30          # FCPP_OUTPUT_PITCH (include/fcpp.h)

c_double_p = C.POINTER(C.c_double)
c_i64_p = C.POINTER(C.c_int64)


class FcppError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f'libfcpp error {code}: {msg}')
        self.code = code


class Vehicle(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        'working_width', 'min_turn_radius', 'max_work_speed_kmh', 'max_headland_speed_kmh',
        'headland_turn_speed_kmh', 'max_lateral_accel', 'max_longitudinal_accel', 'safety_factor')]


class Options(C.Structure):
    _fields_ = [('turn_model', C.c_int32), ('clothoid_fit', C.c_int32), ('sample_spacing', C.c_double),
                ('clothoid_frac', C.c_double), ('geofence_tol', C.c_double), ('obstacle_mode', C.c_int32), ('ring_order', C.c_int32)]


class Field(C.Structure):
    _fields_ = [('vx', C.c_double * 4), ('vy', C.c_double * 4), ('from_vertices', C.c_int32),
                ('has_start', C.c_int32), ('has_end', C.c_int32),
                ('start_x', C.c_double), ('start_y', C.c_double), ('end_x', C.c_double), ('end_y', C.c_double),
                ('n_obstacles', C.c_int32), ('_pad', C.c_int32), ('obstacle_first', C.c_int64)]


class Polys(C.Structure):
    _fields_ = [('n_polys', C.c_int64), ('offsets', c_i64_p), ('x', c_double_p), ('y', c_double_p)]


class FieldInfo(C.Structure):
    _fields_ = [('point_offset', C.c_int64), ('n_main', C.c_int64), ('n_head', C.c_int64),
                ('n_swaths', C.c_int32), ('n_loops', C.c_int32), ('start_corner', C.c_int32),
                ('reverse_order', C.c_int32), ('start_from_right', C.c_int32), ('rotated', C.c_int32),
                ('start_kept', C.c_int32), ('end_kept', C.c_int32), ('shape', C.c_int32),
                ('n_reverse', C.c_int32 * 4), ('status', C.c_int32),
                ('corner_angles', C.c_double * 4), ('field_length', C.c_double), ('field_width', C.c_double),
                ('headland_width', C.c_double), ('rotation_angle', C.c_double),
                ('approach_from', C.c_double * 2), ('approach_to', C.c_double * 2),
                ('departure_from', C.c_double * 2), ('departure_to', C.c_double * 2)]


class FieldStats(C.Structure):
    _fields_ = [('main_len_m', C.c_double), ('main_time_pre_s', C.c_double), ('main_time_s', C.c_double),
                ('head_len_m', C.c_double), ('head_time_pre_s', C.c_double), ('head_time_s', C.c_double),
                ('max_kappa', C.c_double), ('max_alat', C.c_double), ('max_jump', C.c_double),
                ('n_viol', C.c_int64), ('n_outside', C.c_int64), ('n_in_obstacle', C.c_int64),
                ('n_adjusted', C.c_int64)]


class SetupTimes(C.Structure):
    """fcpp_setup_times: where the time of fcpp_batch_create went"""
    _fields_ = [('host_plan_ms', C.c_double), ('templates_ms', C.c_double), ('tiler_ms', C.c_double), ('image_ms', C.c_double),
                ('h2d_ms', C.c_double), ('total_ms', C.c_double), ('image_bytes', C.c_int64), ('threads', C.c_int32), ('device_setup', C.c_int32)]


class GaConfig(C.Structure):
    """fcpp_ga_config = GAConfig (GA:20-29) + seed"""
    _fields_ = [('population_size', C.c_int32), ('max_generations', C.c_int32), ('crossover_rate', C.c_double),
                ('mutation_rate', C.c_double), ('elite_size', C.c_int32), ('tournament_size', C.c_int32),
                ('convergence_threshold', C.c_int32), ('_pad', C.c_int32), ('seed', C.c_uint64)]


class GaResult(C.Structure):
    _fields_ = [('generations', C.c_int32), ('convergence_gen', C.c_int32), ('best_distance', C.c_double),
                ('best_fitness', C.c_double)]


class CoverJob(C.Structure):
    """fcpp_cover_job (include/fcpp.h)"""
    _fields_ = [('ox', C.c_double), ('oy', C.c_double), ('res', C.c_double), ('shift', C.c_double), ('radius', C.c_double),
                ('nx', C.c_int32), ('ny', C.c_int32), ('n_a', C.c_int32), ('n_b', C.c_int32),
                ('pts_first', C.c_int64), ('grid_first', C.c_int64), ('strict', C.c_int32), ('region', C.c_int32),
                ('outer', C.c_double * 12), ('inner', C.c_double * 12)]


SETUP_AUTO, SETUP_HOST, SETUP_DEVICE = 0, 1, 2      # fcpp_ctx_set_setup

STATS_DOUBLES = 9   # leading float64 members of FieldStats
STATS_WORDS = 13    # 8-byte words per FieldStats

# every symbol include/fcpp.h declares: (name, restype, argtypes)
_VP = C.c_void_p
PROTOTYPES = [
    ('fcpp_last_error', C.c_char_p, []),
    ('fcpp_abi_version', C.c_int, []),
    ('fcpp_vehicle_default', None, [C.POINTER(Vehicle)]),
    ('fcpp_options_default', None, [C.POINTER(Options)]),
    ('fcpp_ctx_create', C.c_int, [C.c_int, C.POINTER(_VP)]),
    ('fcpp_ctx_destroy', C.c_int, [_VP]),
    ('fcpp_ctx_set_stream', C.c_int, [_VP, _VP]),
    ('fcpp_ctx_synchronize', C.c_int, [_VP]),
    ('fcpp_ctx_set_setup', C.c_int, [_VP, C.c_int]),
    ('fcpp_malloc', C.c_int, [_VP, C.c_int64, C.POINTER(_VP)]),
    ('fcpp_free', C.c_int, [_VP, _VP]),
    ('fcpp_ctx_reserve_outputs', C.c_int, [_VP, C.c_int64, C.c_int64]),
    ('fcpp_ctx_outputs_info', C.c_int, [_VP, c_i64_p, c_i64_p, c_i64_p]),
    ('fcpp_outputs_alloc', C.c_int, [_VP, C.c_int64, C.c_int64] + [C.POINTER(_VP)] * 5),
    ('fcpp_outputs_free', C.c_int, [_VP, _VP]),
    ('fcpp_memcpy_h2d', C.c_int, [_VP, _VP, _VP, C.c_int64]),
    ('fcpp_memcpy_d2h', C.c_int, [_VP, _VP, _VP, C.c_int64]),
    ('fcpp_plan_count', C.c_int, [C.POINTER(Vehicle), C.POINTER(Options), C.c_int64, C.POINTER(Field), C.POINTER(Polys),
                                  C.POINTER(FieldInfo)]),
    ('fcpp_plan_points', C.c_int, [_VP, C.POINTER(Vehicle), C.POINTER(Options), C.c_int64, C.POINTER(Field), C.POINTER(Polys), c_i64_p]),
    ('fcpp_batch_create', C.c_int, [_VP, C.POINTER(Vehicle), C.POINTER(Options), C.c_int64, C.POINTER(Field),
                                    C.POINTER(Polys), C.POINTER(_VP)]),
    ('fcpp_batch_plan', C.c_int, [_VP, C.POINTER(Vehicle), C.POINTER(Options), C.c_int64, C.POINTER(Field), C.POINTER(Polys), _VP,
                                  C.POINTER(_VP)] + [C.POINTER(_VP)] * 5 + [c_i64_p]),
    ('fcpp_batch_own_stats', C.c_int, [_VP, C.POINTER(_VP)]),
    ('fcpp_batch_info', C.c_int, [_VP, C.POINTER(FieldInfo), c_i64_p]),
    ('fcpp_batch_setup_times', C.c_int, [_VP, C.POINTER(SetupTimes)]),
    ('fcpp_batch_run', C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_int]),
    ('fcpp_batch_connectors', C.c_int, [_VP, _VP, _VP]),
    ('fcpp_batch_destroy', C.c_int, [_VP]),
    ('fcpp_batch_set_profiling', C.c_int, [_VP, C.c_int]),
    ('fcpp_batch_stage_times', C.c_int, [_VP, C.c_int, c_double_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ('fcpp_batch_stage_name', C.c_char_p, [C.c_int, C.c_int]),
    ('fcpp_batch_stage_points', C.c_int, [_VP, C.c_int, C.c_int, c_i64_p]),
    ('fcpp_batch_point_split', C.c_int, [_VP, c_i64_p, c_i64_p]),
    ('fcpp_batch_reduce_classes', C.c_int, [_VP, c_i64_p]),
    ('fcpp_curvature', C.c_int, [_VP, C.c_int64, _VP, C.c_int64, _VP, _VP, _VP, _VP]),
    ('fcpp_speed_plan', C.c_int, [_VP, C.POINTER(Vehicle), C.c_int, C.c_int64, _VP, C.c_int64, _VP, _VP, _VP, _VP,
                                  _VP, _VP, _VP]),
    ('fcpp_verify', C.c_int, [_VP, C.POINTER(Vehicle), C.c_int64, _VP, C.c_int64, _VP, _VP, _VP, _VP, _VP]),
    ('fcpp_validate', C.c_int, [_VP, C.POINTER(Vehicle), C.POINTER(Options), C.c_int64, _VP, C.c_int64, _VP, _VP, _VP, C.POINTER(Polys),
                               C.POINTER(Polys), _VP, _VP, _VP, _VP]),
    ('fcpp_straight_segments', C.c_int, [_VP, C.c_int64, _VP, C.c_int32, _VP]),
    ('fcpp_corner_turns', C.c_int, [_VP, C.POINTER(Vehicle), C.c_int64, _VP, _VP, _VP, C.c_double, C.c_double, C.c_int32, _VP, _VP]),
    ('fcpp_fresnel', C.c_int, [_VP, C.c_int64, _VP, _VP, _VP]),
    ('fcpp_ga_fitness', C.c_int, [_VP, C.c_int32, C.c_int64, _VP, _VP, _VP, _VP, C.c_int]),
    ('fcpp_distance_matrix', C.c_int, [_VP, C.c_int32, _VP, _VP, _VP]),
    ('fcpp_best_connections', C.c_int, [_VP, C.c_int64, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    ('fcpp_ga_evolve', C.c_int, [_VP, C.c_int32, C.POINTER(GaConfig), _VP, _VP, _VP, _VP, C.POINTER(GaResult)]),
    ('fcpp_cover_grid', C.c_int, [_VP, C.c_int64, C.POINTER(CoverJob), C.c_int64, _VP, _VP, _VP, _VP]),
    ('fcpp_gather', C.c_int, [_VP, _VP, C.c_int, C.c_int, C.c_int, C.c_int, _VP, _VP, c_i64_p, _VP, C.c_int]),
    ('fcpp_debug_math', C.c_int, [C.c_int, C.c_int64, _VP, _VP, _VP, _VP]),
    ('fcpp_debug_math_dev', C.c_int, [_VP, C.c_int, C.c_int64, _VP, _VP, _VP, _VP]),
    ('fcpp_batch_debug_table', C.c_int, [_VP, C.c_int, _VP, C.c_int64, c_i64_p]),
]

_lib = None


def load():
    """Load libfcpp.so (once).  Raises if the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                f'or `make -C {os.path.join(_HERE, "csrc")}`; there is no CPU fallback')
        # One HIP runtime per process: PyTorch's ROCm wheels bundle their own libamdhip64 (same SONAME as the system's).
        # Imported first, torch's copy is the one libfcpp's DT_NEEDED resolves to, so streams and allocations are shared;
        # loaded the other way round the process would end up with two runtimes that cannot see each other's streams.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(LIB_PATH)
        for name, res, args in PROTOTYPES:
            fn = getattr(lib, name)   # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        if lib.fcpp_abi_version() != 5:
            raise ImportError('libfcpp.so ABI version mismatch')
        _lib = lib
    return _lib


def check(rc):
    if rc != OK:
        raise FcppError(rc, load().fcpp_last_error().decode('utf-8', 'replace'))


def default_vehicle():
    v = Vehicle()
    load().fcpp_vehicle_default(C.byref(v))
    return v


def default_options():
    o = Options()
    load().fcpp_options_default(C.byref(o))
    return o
