"""ctypes binding of libintel_hip.so (include/intel_hip.h).

The product path has NO CPU fallback: if the library is missing or a call fails, an exception
is raised (IntelHipError).  PyTorch is used only for device memory and streams.
"""
import ctypes as C
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libintel_hip.so')


class IntelHipError(RuntimeError):
    pass


_lib = None


def lib():
    """Load the library (once).  Raises IntelHipError when it has not been built."""
    global _lib
    if _lib is None:
        from . import build as _build
        if not _build.is_current():
            # a fresh checkout or edited csrc/: compile in-tree (hipcc cross-compiles without a GPU; build_library serialises
            # concurrent ranks).  A stale library is never loaded silently: if it cannot be rebuilt this raises.
            try:
                _build.build_library()
            except Exception as e:
                raise IntelHipError('libintel_hip.so is missing or older than csrc/ and could not be rebuilt (%s): run `python -c '
                                    '"import __graft_entry__ as g; g.build()"`; there is no CPU fallback' % e)
        try:
            _lib = C.CDLL(LIB_PATH)
        except OSError as e:
            raise IntelHipError('cannot load %s: %s' % (LIB_PATH, e))
        _declare(_lib)
        if _lib.intel_abi_version() != 5:
            raise IntelHipError('ABI version mismatch')
        sizes = (C.c_int * 4)()
        _lib.intel_abi_sizes(sizes)
        mine = [C.sizeof(IntelDesc), C.sizeof(IntelBatch), C.sizeof(IntelOut), P_COUNT]
        if list(sizes) != mine:
            raise IntelHipError('struct layout mismatch between _lib.py and intel_hip.h: %s vs %s' % (list(sizes), mine))
        fs = (C.c_int * 2)()
        _lib.intel_feed_abi_sizes(fs)
        if list(fs) != [C.sizeof(IntelFeedStore), C.sizeof(IntelFeedOut)]:
            raise IntelHipError('feed struct layout mismatch between _lib.py and intel_hip.h')
        if _lib.intel_lazy_table_sizeof() != C.sizeof(IntelLazyTable):
            raise IntelHipError('IntelLazyTable layout mismatch between _lib.py and intel_hip.h')
    return _lib


def check(rc, what=''):
    if rc != 0:
        msg = lib().intel_last_error().decode('utf-8', 'replace')
        raise IntelHipError('%s failed (code %d): %s' % (what or 'intel_hip call', rc, msg))


def ptr(t):
    """Device pointer of a tensor (or None)."""
    if t is None:
        return None
    assert t.is_contiguous(), 'tensor must be contiguous'
    return C.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_gpu(t):
    if not t.is_cuda:
        raise IntelHipError('intel_sigir2023_amd runs on MI355X only (tensor on %s); there is no CPU path' % t.device)


# ---- struct mirrors of include/intel_hip.h ------------------------------------------------------
class IntelDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        'model_num', 'intent_num', 'item_num', 'class_num', 'user_num', 'ctx_num',
        'd_id', 'd_im', 'd_u', 'd_s', 'd_c', 'd_int', 'q_size', 'heads', 'layers',
        'cross_attention', 'encoder', 'history_max', 'enc_layers', 'enc_heads', 'gru_hidden', 'weight_norm', 'pool_mean', 'dtype')]


class IntelBatch(C.Structure):
    _fields_ = [('B', C.c_int), ('L', C.c_int), ('H', C.c_int), ('Hi', C.c_int)] + [(n, C.c_void_p) for n in (
        'i_id_s', 'i_class_c', 'scores', 'session_len', 'u_id_c', 'context_mh', 'his_context_mh',
        'his_intents', 'history_len', 'his_item_id', 'his_item_idx', 'his_item_int', 'history_item_len', 'his_off', 'hisitem_off')] + \
        [('n_his_rows', C.c_int), ('n_hisitem_rows', C.c_int)] + \
        [(n, C.c_void_p) for n in ('iid_sort_ids', 'iid_sort_rows', 'cls_sort_ids', 'cls_sort_rows', 'hisitem_sort_ids', 'hisitem_sort_rows', 'his_order', 'hisitem_order')]


class IntelFeedStore(C.Structure):
    _fields_ = [(n, C.c_int) for n in ('n_sessions', 'n_users', 'n_scores', 'intent_num', 'max_his', 'n_intent_rows')] + \
               [(n, C.c_void_p) for n in ('u_id', 'context_mh', 'n_pay', 'n_fav', 'n_click', 'n_trueneg', 'position', 'item_position',
                                          'intent_row', 'list_off', 'item_id', 'item_class', 'scores', 'intent_rows', 'uhis_off',
                                          'uhis_context_mh', 'uhis_intent_row', 'uitem_off', 'uitem_id', 'uitem_intent_idx')]


class IntelFeedOut(C.Structure):
    _fields_ = [('B', C.c_int), ('L', C.c_int), ('H', C.c_int), ('Hi', C.c_int)] + \
               [(n, C.c_void_p) for n in ('i_id_s', 'i_class_c', 'scores', 'ranking', 'session_len', 'u_id_c', 'context_mh', 'intents',
                                          'his_context_mh', 'his_intents', 'history_len', 'his_item_id', 'his_item_idx', 'history_item_len')]


class IntelLazyTable(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ('p', 'm', 'v', 'last', 'sched')] + [('rows', C.c_longlong), ('d', C.c_int), ('base', C.c_int),
               ('cap', C.c_int), ('beta1', C.c_float), ('beta2', C.c_float), ('eps', C.c_float), ('weight_decay', C.c_float)]


class IntelOut(C.Structure):
    _fields_ = [('weights', C.c_void_p), ('ens_score', C.c_void_p), ('intents', C.c_void_p)]


# parameter slots (enum IntelParam)
_P_NAMES = ['IID_EMB', 'ITEM_EMB', 'UID_EMB', 'CTX_EMB', 'INTENT_W', 'INTENT_B', 'SCORE_W', 'SCORE_B',
            'I_WQ', 'I_WK', 'I_WV', 'I_W1', 'I_B1', 'I_W2', 'I_B2', 'I_LNG', 'I_LNB',
            'S_WQ', 'S_WK', 'S_WV', 'S_W1', 'S_B1', 'S_W2', 'S_B2', 'S_LNG', 'S_LNB',
            'XI_WQ', 'XI_WK', 'XI_WV', 'XS_WQ', 'XS_WK', 'XS_WV',
            'MI_W0', 'MI_B0', 'MI_W2', 'MS_W0', 'MS_B0', 'MS_W2', 'WE_W', 'WE_B', 'PRED_W', 'PRED_B']
P = {n: i for i, n in enumerate(_P_NAMES)}
P_ENC0 = len(_P_NAMES)
ENC_POS, ENC_GRU_WIH, ENC_GRU_WHH, ENC_GRU_BIH, ENC_GRU_BHH, ENC_GRU_OUT, ENC_BLOCK0 = 0, 1, 2, 3, 4, 5, 6
ENC_BLOCK_NAMES = ['WQ', 'BQ', 'WK', 'BK', 'WV', 'BV', 'LN1G', 'LN1B', 'W1', 'B1', 'W2', 'B2', 'LN2G', 'LN2B']
ENC_BLOCK_STRIDE, ENC_MAX_BLOCKS = 14, 4
ENC_STRIDE = 6 + ENC_BLOCK_STRIDE * ENC_MAX_BLOCKS
P_COUNT = P_ENC0 + 2 * ENC_STRIDE

EXPORTS = [
    'intel_last_error', 'intel_abi_version', 'intel_abi_sizes', 'intel_create', 'intel_destroy', 'intel_set_concurrency', 'intel_set_params_unchanged', 'intel_set_table_stream', 'intel_set_table_wait_event', 'intel_side_stream', 'intel_set_dropout', 'intel_set_iid_grad_row_flags', 'intel_workspace_bytes',
    'intel_forward', 'intel_backward', 'intel_backward_phase', 'intel_bpr_loss', 'intel_bpr_loss_seeded', 'intel_list_loss', 'intel_mse_loss', 'intel_intent_loss',
    'intel_loss_workspace_bytes', 'intel_loss_total', 'intel_adam_step', 'intel_adam_step_pair', 'intel_adam_step_rows', 'intel_ndcg', 'intel_eval_metrics', 'intel_op_linear',
    'intel_op_linear_dgrad', 'intel_op_linear_wgrad', 'intel_op_linear_bwd', 'intel_op_linear_bwd_workspace_bytes', 'intel_op_linear_bwd_qkv', 'intel_op_linear_bwd_qkv_workspace_bytes', 'intel_op_attention', 'intel_op_attention_bwd', 'intel_op_attention_bwd_workspace_bytes',
    'intel_op_add_layernorm', 'intel_op_workspace_bytes', 'intel_prof_enable', 'intel_prof_collect', 'intel_prof_timeline', 'intel_feed_collate', 'intel_feed_abi_sizes', 'intel_rows_take', 'intel_rows_add', 'intel_rows_compact', 'intel_rows_compact_scratch_ints', 'intel_rows_mark',
    'intel_lazy_table_sizeof', 'intel_adam_lazy_step', 'intel_adam_lazy_catchup', 'intel_adam_lazy_flush', 'intel_set_lazy_table',
]


def _declare(l):
    vp, i, f, d, sz, ll = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_size_t, C.c_longlong
    l.intel_last_error.restype = C.c_char_p
    l.intel_abi_version.restype = i

    def sig(name, res, args):
        if hasattr(l, name):
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
    sig('intel_abi_sizes', None, [C.POINTER(C.c_int)])
    sig('intel_create', vp, [C.POINTER(IntelDesc)])
    sig('intel_destroy', None, [vp])
    sig('intel_set_concurrency', None, [vp, i])
    sig('intel_set_params_unchanged', None, [vp, i])
    sig('intel_set_table_stream', None, [vp, vp])
    sig('intel_set_table_wait_event', None, [vp, vp])
    sig('intel_side_stream', vp, [vp, i])
    sig('intel_set_dropout', i, [vp, f, C.c_ulonglong, vp])
    sig('intel_set_iid_grad_row_flags', i, [vp, vp])
    sig('intel_workspace_bytes', sz, [vp, i, i, i, i, i])
    sig('intel_forward', i, [vp, C.POINTER(vp), C.POINTER(IntelBatch), vp, sz, C.POINTER(IntelOut), i, vp])
    sig('intel_backward', i, [vp, C.POINTER(vp), C.POINTER(IntelBatch), vp, sz, vp, vp, vp, C.POINTER(vp), vp])
    sig('intel_backward_phase', i, [vp, C.POINTER(vp), C.POINTER(IntelBatch), vp, sz, vp, vp, vp, C.POINTER(vp), i, vp])
    sig('intel_bpr_loss', i, [i, i, i, vp, vp, vp, vp, vp, vp, vp, i, d, f, vp, vp, vp, vp, vp, sz, vp])
    sig('intel_bpr_loss_seeded', i, [i, i, i, vp, vp, vp, C.c_ulonglong, C.c_ulonglong, vp, vp, vp, i, d, f, vp, vp, vp, vp, vp, sz, vp])
    sig('intel_list_loss', i, [i, i, i, vp, vp, vp, vp, vp, vp, i, d, f, vp, vp, vp, vp, sz, vp])
    sig('intel_mse_loss', i, [i, i, i, vp, vp, vp, vp, vp, vp, i, d, f, vp, vp, vp, vp, sz, vp])
    sig('intel_intent_loss', i, [i, i, vp, vp, d, d, f, vp, vp, vp, sz, vp])
    sig('intel_loss_workspace_bytes', sz, [i, i, i])
    sig('intel_loss_total', i, [vp, vp, d, d, vp, vp])
    sig('intel_adam_step', i, [vp, vp, vp, vp, ll, f, f, f, f, f, i, f, i, vp])
    sig('intel_adam_step_rows', i, [vp, vp, vp, vp, ll, i, vp, f, f, f, f, f, i, f, vp])
    sig('intel_adam_step_pair', i, [vp, vp, vp, vp, vp, vp, f, f, f, f, i, f, i, vp])
    sig('intel_lazy_table_sizeof', i, [])
    sig('intel_adam_lazy_step', i, [C.POINTER(IntelLazyTable), vp, vp, f, i, vp])
    sig('intel_adam_lazy_catchup', i, [C.POINTER(IntelLazyTable), vp, ll, vp, ll, i, vp])
    sig('intel_adam_lazy_flush', i, [C.POINTER(IntelLazyTable), i, vp])
    sig('intel_set_lazy_table', i, [vp, C.POINTER(IntelLazyTable), i])
    sig('intel_ndcg', i, [i, i, i, vp, vp, vp, vp, vp])
    sig('intel_eval_metrics', i, [i, i, i, i, C.POINTER(C.c_int), vp, vp, vp, vp, vp, vp, vp, vp])
    sig('intel_op_linear', i, [vp, i, i, vp, i, vp, i, vp, vp, sz, vp])
    sig('intel_op_linear_dgrad', i, [vp, i, i, vp, i, vp, vp, sz, vp])
    sig('intel_op_linear_wgrad', i, [vp, vp, i, i, i, vp, vp, vp, sz, vp])
    sig('intel_op_linear_bwd', i, [vp, vp, i, i, vp, i, vp, vp, vp, vp, sz, vp])
    sig('intel_op_linear_bwd_workspace_bytes', sz, [i, i])
    sig('intel_op_linear_bwd_qkv', i, [vp, vp, vp, i, i, i, vp, vp, vp, vp, vp, sz, vp])
    sig('intel_op_linear_bwd_qkv_workspace_bytes', sz, [i, i, i])
    sig('intel_op_attention', i, [vp, i, i, i, i, vp, vp, vp, vp])
    sig('intel_op_attention_bwd', i, [vp, vp, vp, vp, i, i, i, i, vp, vp, vp, vp])
    sig('intel_op_attention_bwd_workspace_bytes', sz, [i, i, i, i])
    sig('intel_op_add_layernorm', i, [vp, vp, i, i, vp, vp, vp, vp, vp, vp])
    sig('intel_op_workspace_bytes', sz, [i, i, i])
    sig('intel_rows_take', i, [vp, i, vp, i, vp, i, vp])
    sig('intel_rows_add', i, [vp, i, vp, i, vp, vp])
    sig('intel_rows_compact', i, [vp, C.c_longlong, vp, i, vp, vp])
    sig('intel_rows_compact_scratch_ints', C.c_longlong, [C.c_longlong])
    sig('intel_rows_mark', i, [vp, vp, i, vp])
    sig('intel_feed_abi_sizes', None, [C.POINTER(C.c_int)])
    sig('intel_feed_collate', i, [C.POINTER(IntelFeedStore), vp, i, vp, C.c_ulonglong, C.POINTER(IntelFeedOut), vp])
    sig('intel_prof_enable', None, [i])
    sig('intel_prof_collect', C.c_char_p, [])
    sig('intel_prof_timeline', C.c_char_p, [])
