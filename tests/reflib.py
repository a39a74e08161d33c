"""Loader for oracle/_ref/libsf_refslice.so: the ranges of the reference that compile with the standard library alone
(oracle/ref_slices.py).  Test infrastructure only.  Returns None where the file was never built (no checkout)."""
import ctypes as C
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "oracle", "_ref", "libsf_refslice.so")
REF_CLIENT = "/root/reference/StrikeForce-client"
_LIB = []


def lib():
    if not _LIB:
        L = None
        if os.path.exists(PATH):
            L = C.CDLL(PATH)
            I32P, I64P = C.POINTER(C.c_int), C.POINTER(C.c_longlong)
            L.ref_rand.argtypes = [C.c_longlong, C.c_longlong, C.c_int, I32P, I64P]
            L.ref_rand.restype = None
            L.ref_compute_damage.argtypes = [C.c_int, C.c_int]
            L.ref_items.argtypes = [C.c_char_p, I32P]
            L.ref_bullet.argtypes = [I32P, I32P] + [C.c_int] * 5 + [I32P]
            L.ref_bullet.restype = None
            L.ref_character_hit.argtypes = [C.c_int] * 4 + [I32P]
            L.ref_character_hit.restype = None
            L.ref_zombie.argtypes = [C.c_int, I32P, C.c_int, C.c_int, C.c_int, C.c_int, I32P]
            L.ref_zombie.restype = None
        _LIB.append(L)
    return _LIB[0]


def rand(tb, serial, n):
    """(first n outputs of _rand() after _srand(tb, serial), [random[0..17], jomle] after them)"""
    out = (C.c_int * max(n, 1))()
    st = (C.c_longlong * 19)()
    lib().ref_rand(tb, serial, n, out, st)
    return list(out)[:n], list(st)


def items():
    out = (C.c_int * 108)()
    if lib().ref_items(REF_CLIENT.encode(), out) != 0:
        return None
    v = list(out)
    cons = [v[6 * i:6 * i + 6] for i in range(4)]
    thr = [v[24 + 7 * i:24 + 7 * i + 7] for i in range(4)]
    w = [v[52 + 7 * i:52 + 7 * i + 7] for i in range(8)]
    return {"cons": cons, "throw": thr, "weapon": w}


def bullet(cor0, cor1, way, damage, effect, range_, owner):
    out = (C.c_int * 6)()
    lib().ref_bullet((C.c_int * 3)(*cor0), (C.c_int * 3)(*cor1), way, damage, effect, range_, owner, out)
    return list(out)


def character_hit(hp, md, damage, effect):
    out = (C.c_int * 2)()
    lib().ref_character_hit(hp, md, damage, effect, out)
    return list(out)


def zombie(super_, cor, hits, damage, effect, way):
    out = (C.c_int * 11)()
    lib().ref_zombie(super_, (C.c_int * 3)(*cor), hits, damage, effect, way, out)
    return list(out)
