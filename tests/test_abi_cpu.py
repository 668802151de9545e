"""The C-ABI library loads on a CPU-only box and exports every symbol include/*.h declares.
No compute calls here (there is no GPU); the product path must refuse to run without one."""
import ctypes as C
import os
import re

import pytest

from ra_amd import abi, engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header="ra_gpu_batch.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    # `static inline` helpers (rgb_decision_expand) are header code, not exports
    inline = set(re.findall(r"static\s+inline\s+[a-z_0-9 ]*?\b(rgb_[a-z0-9_]+)\s*\(", src))
    return sorted(set(re.findall(r"\b(rgb_[a-z0-9_]+)\s*\(", src)) - inline)


@pytest.fixture(scope="module")
def L():
    engine.build()
    return engine.lib()


def test_header_functions_are_all_exported(L):
    names = declared_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(L, n), f"libra_gpu_batch.so does not export {n}"
    assert sorted(engine.EXPORTS) == names
    # every header under include/ is covered: the load generator and the WAL checksum entry points
    assert sorted(os.listdir(os.path.join(ROOT, "include"))) == ["ra_gpu_batch.h", "ra_gpu_batch_synth.h",
                                                                  "ra_gpu_wal.h"]
    for header, listed in (("ra_gpu_batch_synth.h", engine.SYNTH_EXPORTS), ("ra_gpu_wal.h", engine.WAL_EXPORTS)):
        names = declared_functions(header)
        assert sorted(listed) == names, header
        for n in names:
            assert hasattr(L, n), f"libra_gpu_batch.so does not export {n} ({header})"


def test_struct_sizes_match_numpy_mirrors(L):
    for i, dt in enumerate(abi.STRUCT_DTYPES):
        assert L.rgb_struct_size(i) == dt.itemsize
    assert abi.MSG_DTYPE.itemsize == 64 and abi.DECISION_DTYPE.itemsize == 64


def test_strerror_and_default_config(L):
    assert L.rgb_strerror(0) == b"ok"
    assert b"no CPU fallback" in L.rgb_strerror(abi.E_NODEVICE)
    cfg = engine.default_config()
    assert int(cfg["abi_version"][0]) == abi.ABI_VERSION
    assert int(cfg["max_pipeline_count"][0]) == 4096      # src/ra_server.hrl:8
    assert int(cfg["max_aer_batch"][0]) == 128            # src/ra_server.hrl:7


def test_open_without_gpu_fails_loudly(L):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    rc = L.rgb_open(None, C.byref(h))
    assert rc == abi.E_NODEVICE and not h.value
    with pytest.raises(engine.RgbError):
        engine.RaGpuBatch(4, 3)


def test_product_does_not_reference_the_oracle():
    """The product path (ra_amd/, include/) must never import, link or call oracle/."""
    bad = []
    for base in ("ra_amd", "include"):
        for dp, _dn, fns in os.walk(os.path.join(ROOT, base)):
            for fn in fns:
                if fn.endswith((".py", ".hip", ".h", ".c", ".cpp", "Makefile")):
                    txt = open(os.path.join(dp, fn), errors="ignore").read()
                    if re.search(r"ra_oracle|libra_oracle|from oracle|import oracle|ora_step", txt):
                        bad.append(os.path.join(dp, fn))
    assert not bad, bad


def test_erlang_stub_constants_agree_with_the_abi():
    """erlang/ra_gpu_batch.erl (the reference-side binding, never compiled here) carries its own copies of
    the message kinds and decision flags: they must be the numbers of include/ra_gpu_batch.h."""
    import os
    import re
    from ra_amd import abi
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "erlang",
                            "ra_gpu_batch.erl")).read()
    defs = {}
    for name, val in re.findall(r"^-define\((\w+),\s*([^)]+)\)\.", src, flags=re.M):
        val = val.strip()
        defs[name] = int(val[3:], 16) if val.startswith("16#") else int(val)
    checked = 0
    for name, val in defs.items():
        if name.startswith(("MSG_", "F_")):
            assert getattr(abi, name) == val, name
            checked += 1
    assert defs["NONE"] == abi.NONE and defs["UNDEF"] == abi.UNDEF_INT
    assert checked >= 25
    # every struct the stub packs by hand has the size the header says
    assert "0:(7 * 64)" in src                                   # 64-byte rgb_msg: 8 header bytes + 7 words
