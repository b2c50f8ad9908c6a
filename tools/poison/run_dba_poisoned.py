"""Runs the DELTA_BYTE_ARRAY parity test with LDS and private memory poisoned before every decode call."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tests.gpu_util as gu
from tests import test_gpu_strings as t

lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "liblds_poison.so"))
word = int(sys.argv[1], 0)
orig = gu.run_strings

def wrapped(*a, **k):
    gu.ctx().sync()
    rc = lib.lds_poison(ctypes.c_uint32(word))
    assert rc == 0, rc
    return orig(*a, **k)

gu.run_strings = wrapped
t.run_strings = wrapped
t.test_dba()
t.test_dlba()
print("poison", hex(word), "ok")
