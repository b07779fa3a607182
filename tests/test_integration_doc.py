"""INTEGRATION.md §B is the binding a reference maintainer copies.  These tests keep it honest (the
round-2 example declared 14 of the 24 pointers of kx_decoder_layer and 13 of the 17 fields of kx_decoder_weights; a caller
who followed it made the library walk `layer[i]` with the wrong stride).

* CPU: the code block's two `_fields_` lists are compared with the header's structs name by name, and a deliberately stale
  binding is refused with KX_ERR_INVALID_ARG ("stale binding") before anything is launched.
* GPU: the block is exec'd as written — raw ctypes, nothing from kosmosx._hip — and its pack() + decoder_forward() run the
  tiny decoder against the CPU oracle.
"""
import ctypes as C
import re
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
DOC = (ROOT / "INTEGRATION.md").read_text()
HEADER = (ROOT / "include" / "kosmosx_hip.h").read_text()
LIB = ROOT / "kosmos-x_amd" / "kosmosx" / "lib" / "libkosmosx_hip.so"


def _doc_block() -> str:
    m = re.search(r"<!-- binding:decoder:begin -->\s*```python\n(.*?)```\s*<!-- binding:decoder:end -->", DOC, flags=re.S)
    assert m, "INTEGRATION.md lost its executable binding block"
    return m.group(1)


def _header_fields(cname: str) -> list:
    body = re.sub(r"/\*.*?\*/", "", HEADER, flags=re.S)
    m = re.search(r"typedef struct \{([^{}]*)\}\s*" + cname + r"\s*;", body, flags=re.S)
    assert m, cname
    names = []
    for decl in m.group(1).split(";"):
        decl = decl.strip()
        if decl:
            for part in decl.split(","):
                names.append(re.findall(r"([A-Za-z_0-9]+)\s*$", part.replace("*", " ").strip())[0])
    return names


@pytest.fixture(scope="module")
def built():
    import importlib.util
    spec = importlib.util.spec_from_file_location("kx_build", ROOT / "kosmos-x_amd" / "build.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.build(verbose=False)
    return LIB


def _exec_block(lib_path) -> dict:
    ns = {"LIB_PATH": str(lib_path)}
    exec(compile(_doc_block(), "INTEGRATION.md#B", "exec"), ns)
    return ns


def test_doc_binding_matches_the_header_field_for_field(built):
    """The block's declarations execute without a GPU (CDLL + struct classes + its own sizeof asserts against
    kx_struct_bytes) and list exactly the header's fields, in order."""
    ns = _exec_block(built)
    assert [f[0] for f in ns["KxDecoderLayer"]._fields_] == _header_fields("kx_decoder_layer")
    assert [f[0] for f in ns["KxDecoderWeights"]._fields_] == _header_fields("kx_decoder_weights")
    from kosmosx import _hip
    assert C.sizeof(ns["KxDecoderLayer"]) == C.sizeof(_hip.DecoderLayer)
    assert C.sizeof(ns["KxDecoderWeights"]) == C.sizeof(_hip.DecoderWeights)
    for (n1, t1), (n2, t2) in zip(ns["KxDecoderWeights"]._fields_, _hip.DecoderWeights._fields_):
        assert n1 == n2 and C.sizeof(t1) == C.sizeof(t2), (n1, n2)
    assert f"kx_version() == {_hip.ABI_VERSION}" in _doc_block()


def test_library_reports_its_struct_sizes(built):
    from kosmosx import _hip
    lib = _hip.load()
    for sid, cls in enumerate(_hip.STRUCT_IDS):
        assert lib.kx_struct_bytes(sid) == C.sizeof(cls) > 0, cls.__name__
    assert lib.kx_struct_bytes(len(_hip.STRUCT_IDS)) == 0 and lib.kx_struct_bytes(-1) == 0
    ids = re.search(r"typedef enum \{([^{}]*)\}\s*kx_struct_id", HEADER, flags=re.S).group(1)
    assert len(re.findall(r"KX_STRUCT_[A-Z_]+\s*=", ids)) == len(_hip.STRUCT_IDS) + 1        # + KX_STRUCT_COUNT


def test_stale_binding_is_refused_not_walked(built):
    """Round 2's documented structs, verbatim in shape: 14 pointers per layer, no size fields.  Every stage entry point
    must return KX_ERR_INVALID_ARG with a message that says what to do — before reading a single layer pointer."""
    from kosmosx import _hip
    lib = _hip.load()

    class OldLayer(C.Structure):
        _fields_ = [(n, C.c_void_p) for n in ("sa_g", "sa_b", "wqkv", "bqkv", "wo", "bo", "wo_colsum",
                                              "fl_g", "fl_b", "w1", "b1", "w2", "b2", "w2_colsum")]

    class OldWeights(C.Structure):
        _fields_ = [("layers", C.c_int32), ("dim", C.c_int32), ("heads", C.c_int32), ("ffn", C.c_int32), ("vocab", C.c_int32),
                    ("act", C.c_int32), ("subln", C.c_int32), ("xpos", C.c_int32), ("eps", C.c_float),
                    ("layer", C.POINTER(OldLayer)), ("ln_g", C.c_void_p), ("ln_b", C.c_void_p), ("wout", C.c_void_p)]
    L = (OldLayer * 2)()
    w = OldWeights(2, 256, 4, 512, 1002, 1, 1, 1, 1e-5, C.cast(L, C.POINTER(OldLayer)), 256, 256, 256)
    pw = C.cast(C.byref(w), C.POINTER(_hip.DecoderWeights))
    assert lib.kx_decoder_workspace_bytes(pw, 1, 8, 0) == 0 and "stale binding" in _hip.last_error()
    rc = lib.kx_decoder_forward(pw, 256, 1, 8, 256, 256, 256, 256, 256, 0, 256, 1 << 20, 0, None)
    assert rc == 1 and "stale binding" in _hip.last_error() and "kosmosx_hip.h" in _hip.last_error()
    rc = lib.kx_decoder_decode_step(pw, 256, 1, 3, 256, 256, 256, 256, 256, 256, 8, 256, 0, 256, 1 << 20, 0, None)
    assert rc == 1 and "stale binding" in _hip.last_error()
    # a current mirror with one size wrong (a layer struct that fell behind) is refused the same way
    good = _hip.DecoderWeights()
    good.layer_bytes -= 8
    assert lib.kx_decoder_forward(C.byref(good), 256, 1, 8, 256, 256, 256, 256, 256, 0, 256, 1 << 20, 0, None) == 1
    assert "stale binding" in _hip.last_error()
    v = _hip.VitWeights()
    v.struct_bytes = 0
    assert lib.kx_vit_forward(C.byref(v), 256, 1, 256, 256, 1 << 20, 0, None) == 1 and "stale binding" in _hip.last_error()
    p = _hip.PerceiverWeights()
    p.layer_bytes = 8
    assert lib.kx_perceiver_forward(C.byref(p), 256, 1, 17, 256, None, 256, 1 << 20, 0, None) == 1
    assert "stale binding" in _hip.last_error()


@pytest.mark.gpu
@pytest.mark.parametrize("layers,dim,ffn,heads,T", [(2, 256, 512, 4, 24), (3, 512, 1024, 8, 130)])
def test_doc_binding_runs_the_decoder_against_the_oracle(built, layers, dim, ffn, heads, T):
    """pack() + decoder_forward() exactly as INTEGRATION.md §B prints them, on a torchscale-shaped decoder (the product's
    Decoder keeps torchscale's attribute tree: layers[i].self_attn.q_proj.A ...), against the CPU oracle: KX_PREC_F16C — the
    arithmetic a parity-bound caller binds — inside the north star's 1e-3, KX_PREC_BF16 at that mode's
    own distance; both bit-equal to the product's own binding of the same library."""
    from helpers import oracle_weights, rel_err
    from kosmosx.model import KosmosLanguage
    from oracle import kosmos_oracle as O
    ns = _exec_block(built)
    lm = KosmosLanguage(vocab_size=1002, dim=dim, depth=layers, ffn_dim=ffn, decoder_heads=heads, _seed=11, _perturb=0.1,
                        _max_positions=256).eval()
    cfg = O.DecoderCfg(layers=layers, dim=dim, ffn=ffn, heads=heads, vocab=1002, max_pos=256)
    tok = torch.randint(2, 1002, (2, T), generator=torch.Generator().manual_seed(5))
    w_or = oracle_weights(lm)
    x_in, _ = O.forward_embedding_tokens(w_or, tok, cfg)                       # what model.py:238-244 hands to the decoder
    ref = O.decoder_forward(w_or, x_in.clone(), cfg, O.Switches())
    ref16 = O.decoder_forward(w_or, x_in.clone(), cfg, O.Switches(emulate_bf16=True))
    lm = lm.to("cuda:0")
    dec = lm.decoder
    xp = dec.layers[0].self_attn.xpos
    tables = [t.to("cuda:0") for t in (*xp.tables(T, 0, False), *xp.tables(T, 0, True))]
    # --- the parity mode: f16c operand rows packed by the document's own operand() ---
    w, keep = ns["pack"](dec, ns["KX_PREC_F16C"])
    assert w.struct_bytes == C.sizeof(ns["KxDecoderWeights"]) and w.layers == layers and w.dim == dim and w.ffn == ffn
    out = ns["decoder_forward"](w, x_in.to("cuda:0"), tables, ns["KX_PREC_F16C"])
    torch.cuda.synchronize()
    assert out.shape == (2, T, 1002) and torch.isfinite(out).all()
    e = rel_err(out, ref)
    print(f"INTEGRATION.md binding, KX_PREC_F16C, {layers}L/{dim}d T={T}: max|d|/rms vs the fp32 oracle = {e:.2e}")
    assert e < 1e-3, e                                   # the north star's tolerance, through the documented binding
    lm.precision = "f16c"
    assert torch.equal(out, dec.run(x_in.to("cuda:0").clone(), "f16c"))       # one library, two bindings: bit for bit
    del w, keep
    # --- the throughput mode the example defaults to ---
    w, keep = ns["pack"](dec)
    out = ns["decoder_forward"](w, x_in.to("cuda:0"), tables)
    torch.cuda.synchronize()
    assert out.shape == (2, T, 1002) and torch.isfinite(out).all()
    assert rel_err(out, ref) < 6e-2                      # bf16 operands against fp32 (the mode's own distance: 3.6e-2 at full size; NOT the tolerance)
    assert rel_err(out, ref16) < 6e-2                    # and against the oracle with emulated operand rounding (not bit-alike)
    lm.precision = "bf16"
    assert torch.equal(out, dec.run(x_in.to("cuda:0").clone(), "bf16"))
