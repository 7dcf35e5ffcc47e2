p='butd_detr_amd/_hiplib.py'
s=open(p).read()
s=s.replace('''ALL_SYMBOLS = dict(POINTNET2_SYMBOLS)
ALL_SYMBOLS.update(ATTENTION_SYMBOLS)''','''_P = _c_void_p
SA_SYMBOLS = {
    "butd_sa_group": (_c_int, [_c_int] * 5 + [_P, _P, _P, _c_long, _P, _c_float, _c_int, _P, _P]),
    "butd_sa_colstats": (_c_int, [_c_long, _c_int, _P, _P, _P, _c_int, _P, _P, _P, _P, _P]),
    "butd_sa_bn_finalize": (_c_int, [_c_int, _c_long, _P, _P, _P, _P, _c_float, _c_float, _c_int]
                            + [_P] * 7 + [_P]),
    "butd_sa_pool_finalize": (_c_int, [_c_int] * 3 + [_P] * 10 + [_P]),
    "butd_sa_pool_bwd_stats": (_c_int, [_c_int] * 3 + [_P] * 8 + [_P]),
    "butd_sa_dz_last": (_c_int, [_c_int] * 4 + [_P] * 11 + [_c_int, _P]),
    "butd_sa_mask_stats": (_c_int, [_c_long, _c_int] + [_P] * 8 + [_P]),
    "butd_sa_dz_mid": (_c_int, [_c_long, _c_int] + [_P] * 8 + [_c_int, _P]),
    "butd_sa_scatter_rows": (_c_int, [_c_int] * 5 + [_P] * 3 + [_P]),
}

ALL_SYMBOLS = dict(POINTNET2_SYMBOLS)
ALL_SYMBOLS.update(ATTENTION_SYMBOLS)
ALL_SYMBOLS.update(SA_SYMBOLS)''')
open(p,'w').write(s)
