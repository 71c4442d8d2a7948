"""Page-locking of caller arrays (`_lib.pin_array`): the budget, least-recently-used eviction, the finalizer, and entries
inherited from another process -- against a stand-in for the library (no device needed: the logic is host-side)."""
import gc
import os

import numpy as np
import pytest

from pytrimal_amd import _lib


class FakeLib:
    def __init__(self):
        self.registered = {}
        self.log = []

    def msa_host_register(self, p, nbytes):
        self.registered[p.value] = nbytes
        self.log.append(("reg", p.value))
        return 0

    def msa_host_unregister(self, p):
        assert p.value in self.registered, "unregister of memory that is not registered"
        del self.registered[p.value]
        self.log.append(("unreg", p.value))
        return 0


@pytest.fixture
def fake(monkeypatch):
    lib = FakeLib()
    monkeypatch.setattr(_lib, "load", lambda: lib)
    monkeypatch.setattr(_lib, "_pinned", type(_lib._pinned)())
    monkeypatch.setenv("PYTRIMAL_AMD_PIN_MB", "4")
    return lib


def test_pin_reupload_and_release_on_collection(fake):
    a = np.zeros((1000, 1000), dtype=np.uint8)
    address = a.ctypes.data
    assert _lib.pin_array(a) and _lib.pin_array(a)  # the second call finds it registered
    assert fake.log == [("reg", address)]
    assert _lib.pinned_bytes() == a.nbytes
    del a
    gc.collect()
    assert fake.log[-1] == ("unreg", address) and not fake.registered
    assert _lib.pinned_bytes() == 0


def test_budget_evicts_least_recently_used(fake):
    arrays = [np.zeros((1500, 1000), dtype=np.uint8) for _ in range(3)]  # 1.5 MB each, budget 4 MB
    assert _lib.pin_array(arrays[0]) and _lib.pin_array(arrays[1])
    assert _lib.pin_array(arrays[0])  # touched: arrays[1] is now the least recently used
    assert _lib.pin_array(arrays[2])
    assert set(fake.registered) == {arrays[0].ctypes.data, arrays[2].ctypes.data}
    assert _lib.pinned_bytes() <= _lib.pin_budget_bytes()
    assert _lib.pin_array(arrays[1])  # pinned again on its next upload, at the expense of arrays[0]
    assert set(fake.registered) == {arrays[2].ctypes.data, arrays[1].ctypes.data}
    del arrays
    gc.collect()
    assert not fake.registered


def test_larger_than_budget_and_opt_out(fake, monkeypatch):
    assert not _lib.pin_array(np.zeros(5 << 20, dtype=np.uint8))
    monkeypatch.setenv("PYTRIMAL_AMD_PIN_MB", "0")
    assert not _lib.pin_array(np.zeros(1000, dtype=np.uint8))
    assert not fake.log


def test_entry_of_another_process_is_not_unregistered_here(fake, monkeypatch):
    a = np.zeros((100, 100), dtype=np.uint8)
    assert _lib.pin_array(a)
    real = os.getpid()
    monkeypatch.setattr(os, "getpid", lambda: real + 1)  # "the child of a fork"
    assert _lib.pinned_bytes() == 0
    assert _lib.pin_array(a)  # registered afresh in this process
    assert fake.log == [("reg", a.ctypes.data), ("reg", a.ctypes.data)]
    monkeypatch.setattr(os, "getpid", lambda: real + 2)  # collected in yet another process: no call into its runtime
    del a
    gc.collect()
    assert [e for e in fake.log if e[0] == "unreg"] == []
