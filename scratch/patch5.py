p='butd_detr_amd/_hiplib.py'
s=open(p).read()
s=s.replace('''ALL_SYMBOLS = dict(POINTNET2_SYMBOLS)
''','''_c_long, _c_u32 = ctypes.c_long, ctypes.c_uint32


class GemmProblem(ctypes.Structure):
    """ctypes mirror of ``butd_gemm_problem`` (include/butd_attention.h)."""
    _fields_ = [("a", _c_void_p), ("a2", _c_void_p), ("b", _c_void_p), ("bias", _c_void_p),
                ("c", _c_void_p), ("bias_grad", _c_void_p),
                ("M", _c_int), ("N", _c_int), ("K", _c_int),
                ("lda_m", _c_long), ("lda_k", _c_long), ("ldb_n", _c_long), ("ldb_k", _c_long),
                ("ldc", _c_long),
                ("scale", _c_float), ("a2_mode", _c_int), ("a2_scale", _c_float),
                ("relu", _c_int), ("accumulate", _c_int), ("ones_col", _c_int), ("split_k", _c_int),
                ("dropout_p", _c_float), ("dropout_site", _c_u32)]


ATTENTION_SYMBOLS = {
    "butd_gemm_grouped": (_c_int, [ctypes.POINTER(GemmProblem), _c_int, _c_void_p, _c_void_p]),
    "butd_attention_fwd": (_c_int, [_c_int] * 5 + [_c_void_p] * 6 + [_c_float, _c_u32, _c_void_p, _c_void_p]),
    "butd_attention_bwd": (_c_int, [_c_int] * 5 + [_c_void_p] * 11 + [_c_float, _c_u32, _c_void_p, _c_void_p]),
    "butd_add_dropout_layernorm_fwd": (_c_int, [_c_int, _c_int] + [_c_void_p] * 4 + [_c_float] + [_c_void_p] * 3
                                       + [_c_float, _c_u32, _c_void_p, _c_void_p]),
    "butd_add_dropout_layernorm_bwd": (_c_int, [_c_int, _c_int] + [_c_void_p] * 10
                                       + [_c_float, _c_u32, _c_void_p, _c_void_p]),
}

ALL_SYMBOLS = dict(POINTNET2_SYMBOLS)
ALL_SYMBOLS.update(ATTENTION_SYMBOLS)
''')
open(p,'w').write(s)
