"""ctypes binding of libkge_mi355.so (include/kge_mi355.h).

The library is the product; this module only locates it, declares its signatures and turns its
error codes into exceptions.  There is no fallback: if the shared object is missing the import
fails loudly, and every device entry point raises when no MI355X is usable.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libkge_mi355.so")

TRANSE, TRANSH, TRANSR, TRANSD = 0, 1, 2, 3
NO_EVENT = 1   # kge_stream_wait_emit: nothing was recorded since the previous wait (include/kge_mi355.h KGE_NO_EVENT)
KGE_MAX_TABLES = 4


class KgeError(RuntimeError):
    pass


class ModelDesc(ctypes.Structure):
    """struct kge_model_desc (include/kge_mi355.h)."""
    _fields_ = [("model", ctypes.c_int32), ("negative_rel", ctypes.c_int32),
                ("ent_total", ctypes.c_int64), ("rel_total", ctypes.c_int64),
                ("ent_dim", ctypes.c_int32), ("rel_dim", ctypes.c_int32),
                ("margin", ctypes.c_float), ("reserved", ctypes.c_int32)]


def _declare(L):
    vp, i64, i32, f32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_float
    # (1) Base.so-compatible subset, declared the way Config.py:30-31 declares `sampling`
    L.setInPath.argtypes = [ctypes.c_char_p]
    L.setOutPath.argtypes = [ctypes.c_char_p]
    L.setWorkThreads.argtypes = [i64]
    L.getWorkThreads.restype = i64
    L.setBern.argtypes = [i64]
    for fn in ("getEntityTotal", "getRelationTotal", "getTripleTotal", "getTrainTotal", "getTrainTotal_",
               "getBatchTotal", "getTestTotal", "getValidTotal"):
        getattr(L, fn).restype = i64
    L.sampling.argtypes = [vp, vp, vp, vp, i64, i64, i64]
    # evaluation subset, declared as Config.py:34-39 declares it
    L.getTailBatch.argtypes = [i64, vp, vp, vp]
    L.getHeadBatch.argtypes = [i64, vp, vp, vp]
    L.testTail.argtypes = [i64, vp]
    L.testTail.restype = ctypes.POINTER(ctypes.c_int64 * 8)
    L.testHead.argtypes = [i64, vp]
    L.testHead.restype = ctypes.POINTER(ctypes.c_int64 * 8)
    L.kge_link_prediction.argtypes = [ctypes.POINTER(ModelDesc), ctypes.POINTER(vp), i64, i64, i64, vp, vp]
    # triple classification (Config.py:41-46 -- with all FOUR arguments of test_triple_classification declared)
    L.getValidBatch.argtypes = [vp] * 6
    L.getTestBatch.argtypes = [vp] * 6
    L.getBestThreshold.argtypes = [vp] * 3
    L.test_triple_classification.argtypes = [vp] * 4
    # (2) engine
    L.kge_last_error.restype = ctypes.c_size_t
    L.kge_last_error.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
    L.kge_device_available.restype = ctypes.c_int
    L.kge_version.restype = ctypes.c_char_p
    L.kge_import_train_arrays.argtypes = [i64, i64, i64, vp, vp, vp, i64]
    L.kge_set_option.argtypes = [ctypes.c_char_p, i64]
    L.kge_last_kernel_ms.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_float)]
    L.kge_kernel_ms_mean.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(i64)]
    L.kge_index_copy.restype = i64
    L.kge_index_copy.argtypes = [ctypes.c_char_p, vp, i64]
    L.kge_get_stream_states.argtypes = [vp, i64]
    L.kge_set_stream_states.argtypes = [vp, i64]
    L.kge_sampling_device.argtypes = [vp, vp, vp, i64, i64, i64, i64, i64, i64, ctypes.POINTER(i64), vp]
    L.kge_sampling_attach.argtypes = [vp, vp, vp, i64, i64, i64, i64, i64, i64, ctypes.POINTER(i64), vp]
    L.kge_sampling_flush.argtypes = [vp]
    L.kge_slice_positions.restype = i64
    L.kge_slice_positions.argtypes = [i64, i64, i64, ctypes.POINTER(i64)]
    L.kge_table_shape.argtypes = [ctypes.POINTER(ModelDesc), ctypes.c_int, ctypes.POINTER(i64), ctypes.POINTER(i64)]
    tabs = ctypes.POINTER(vp)
    L.kge_forward_backward.argtypes = [ctypes.POINTER(ModelDesc), tabs, vp, vp, vp, i64, i64, i64, i64, tabs, vp, vp]
    L.kge_stream_wait_emit.argtypes = [vp]
    L.kge_pair_path_active.argtypes = [ctypes.POINTER(ModelDesc), i64, i64]
    L.kge_loss_limbs_target.argtypes = [vp]
    L.kge_forward_backward_sgd_rows.argtypes = [ctypes.POINTER(ModelDesc), tabs, vp, vp, vp, i64, i64, i64, i64, f32, vp, vp]
    L.kge_forward_backward_records.argtypes = [ctypes.POINTER(ModelDesc), tabs, vp, vp, vp, i64, i64, i64, i64, i64, vp, vp, i64, i64, vp, vp]
    L.kge_float_records_apply.argtypes = [ctypes.POINTER(ModelDesc), tabs, vp, vp, i64, i64, i64, f32, vp]
    L.kge_forward_backward_sampled.argtypes = [ctypes.POINTER(ModelDesc), tabs, vp, vp, vp, i64, i64, i64, i64, tabs, vp, vp]
    L.kge_loss_to_limbs.argtypes = [vp, vp, vp]
    L.kge_limbs_to_loss.argtypes = [vp, vp, vp]
    L.kge_sgd_update.argtypes = [vp, vp, i64, f32, vp]
    L.kge_adam_update.argtypes = [vp, vp, vp, vp, i64, f32, f32, f32, f32, vp]
    L.kge_sgd_update_tables.argtypes = [i32, vp, vp, vp, f32, vp]
    L.kge_adam_update_tables.argtypes = [i32, vp, vp, vp, vp, vp, f32, f32, f32, f32, vp]
    L.kge_predict.argtypes = [ctypes.POINTER(ModelDesc), tabs, vp, vp, vp, i64, vp, vp]
    L.kge_transe_counts_supported.argtypes = [ctypes.POINTER(ModelDesc), i64]
    L.kge_transe_forward_counts.argtypes = [ctypes.POINTER(ModelDesc), vp, vp, vp, vp, vp, i64, i64, i64, i64, vp, vp, vp, vp, vp]
    L.kge_transe_apply_counts.argtypes = [vp, vp, vp, vp, vp, i64, i32, i64, i32, f32, f32, f32, f32, vp]
    L.kge_transe_deferred_groups.argtypes = [ctypes.POINTER(ctypes.c_int32)]
    L.kge_transe_apply_counts_tables.argtypes = [ctypes.POINTER(ModelDesc), vp, vp, vp, vp, vp, i64, i32, f32, f32, f32, f32, vp]
    L.kge_transe_train_step_counts.argtypes = [ctypes.POINTER(ModelDesc), vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, vp, vp, i32, i32,
                                               f32, f32, f32, f32, vp, vp]
    L.kge_transe_lazy_row_live.argtypes = [vp]
    L.kge_transe_record_dwords.restype = i64
    L.kge_transe_record_dwords.argtypes = [ctypes.POINTER(ModelDesc)]
    L.kge_transe_emit_records.argtypes = [ctypes.POINTER(ModelDesc), vp, vp, vp, vp, vp, i64, i64, i64, i64, vp, vp, vp, vp, vp, vp]
    L.kge_transe_reduce_records.argtypes = [ctypes.POINTER(ModelDesc), vp, vp, i64, vp, vp, vp, vp]
    L.kge_transe_apply_rows_sgd.argtypes = [ctypes.POINTER(ModelDesc), vp, vp, vp, vp, vp, i64, i64, f32, vp]
    L.kge_transe_apply_rows_adam_lazy.argtypes = [ctypes.POINTER(ModelDesc), vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, f32, f32, f32, f32, vp]
    L.kge_transe_reduce_apply_records_sgd.argtypes = [ctypes.POINTER(ModelDesc), vp, vp, i64, vp, vp, vp, vp, vp, i64, f32, vp]
    L.kge_transe_apply_counts_range.argtypes = [ctypes.POINTER(ModelDesc), vp, vp, vp, vp, vp, i64, i64, i64, i32, f32, f32, f32, f32, vp]
    # table-sharded sparse path (csrc/shard.hip)
    L.kge_shard_requests.argtypes = [vp, vp, vp, i64, i64, i64, vp, vp]
    L.kge_shard_count.argtypes = [vp, i64, i64, i64, vp, vp]
    L.kge_shard_scatter.argtypes = [vp, i64, i64, i64, vp, vp, vp, vp, vp]
    L.kge_shard_remap_batch.argtypes = [vp, vp, i64, i64, i64, vp, vp, vp, vp]
    L.kge_shard_gather_rows.argtypes = [vp, vp, i64, i64, i64, i64, vp, vp]
    L.kge_shard_record_ids.argtypes = [vp, i64, i64, vp, vp, vp]
    L.kge_shard_pack_records.argtypes = [vp, vp, i64, i64, vp, vp]
    L.kge_shard_relation_counts.argtypes = [vp, vp, i64, i64, i64, i64, i64, vp, vp]
    L.kge_shard_scatter_count_rows.argtypes = [vp, vp, vp, i64, i64, i64, i64, vp, vp]
    L.kge_train_steps_persistent.argtypes = [ctypes.POINTER(ModelDesc), vp, vp, vp, vp, i64, i64, i64, i64, i32, vp, f32, f32, f32, vp, vp]
    L.kge_persistent_aborted.argtypes = [ctypes.POINTER(ctypes.c_int32)]
    L.kge_persistent_trace.argtypes = [vp, i64]
    return L


def load(path=None):
    """Load the engine library.  Raises ImportError when it has not been built."""
    path = path or LIB_PATH
    try:
        # PyTorch (the host plumbing: device memory, streams, torch.distributed) bundles its own HIP
        # runtime; it must be the one already resident when the engine's libamdhip64 dependency is
        # resolved, or the process ends up with two runtimes and the engine sees no device.
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(path):
        raise ImportError(
            "%s not found: build it first (python -c 'import __graft_entry__ as g; g.build()' or "
            "make -C openkeonspark_amd/csrc).  There is no CPU fallback." % path)
    return _declare(ctypes.CDLL(path))


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        _LIB = load()
    return _LIB


def last_error(L=None):
    L = L or lib()
    buf = ctypes.create_string_buffer(1024)
    L.kge_last_error(buf, 1024)
    return buf.value.decode(errors="replace")


def check(rc, L=None):
    """Raise KgeError for a negative return code of a kge_* call."""
    if rc is not None and rc < 0:
        raise KgeError("kge_mi355 error %d: %s" % (rc, last_error(L)))
    return rc


def raise_if_error(L=None):
    """For the void Base-compatible calls: raise if they recorded an error."""
    L = L or lib()
    msg = last_error(L)
    if msg:
        L.kge_clear_error()
        raise KgeError(msg)


def table_ptrs(ptrs):
    """list of up to 4 device addresses (or None) -> (void*)[4]"""
    arr = (ctypes.c_void_p * KGE_MAX_TABLES)()
    for i, p in enumerate(ptrs):
        arr[i] = p
    return arr
