"""CPU: the C-ABI library loads and exports every symbol include/tabgnn_hip.h declares; host logic of the product
package (module construction, reference state-dict key names, launch plans, error behaviour) without a GPU."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "tabgnn_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tg_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from tabgnn_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/tabgnn_hip.h but not exported"
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)   # ctypes table mirrors the header
    assert lib.tg_abi_version() == _lib.ABI_VERSION


def test_abi_structs_match_header_layout():
    from tabgnn_amd import _lib
    assert ctypes.sizeof(_lib.EncCol) == 32 and ctypes.sizeof(_lib.EncDesc) == 8 + 16 * 32
    assert ctypes.sizeof(_lib.EncPtrs) == 4 * 16 + 11 * 8          # + row_ids (ABI 2)


def test_product_path_refuses_cpu_tensors():
    """No CPU fallback: the operators fail loudly instead of computing on the host."""
    from tabgnn_amd import ops
    with pytest.raises(RuntimeError, match="no CPU path|MI355X"):
        ops.layer_norm(torch.zeros(4, 32), torch.ones(32), torch.zeros(32))
    with pytest.raises(RuntimeError):
        ops.SubgraphIndex.build(torch.zeros(2, 3, dtype=torch.int64), 4)


def test_state_dict_keys_match_reference_checkpoints():
    import tabgnn_amd as T
    for case, kind in [("fused_c32_h8_l1", "fused"), ("fused_c32_h4_l2_rmp", "fused"), ("tabgnn_c32_h8_l2", "tabgnn")]:
        z = np.load(os.path.join(ROOT, "tests", "golden", case + ".npz"))
        cfg = json.loads(str(z["cfg"]))
        deg = torch.tensor([3, 5, 2, 1])
        C = cfg["C"]
        if kind == "fused":
            m = T.TABGNNFused(channels=C, num_layers=cfg["L"], deg=deg, node_dim=C, nhidden=C, edge_dim=cfg["ncols"] * C,
                              reverse_mp=cfg["reverse_mp"], nhead=cfg["H"])
            h = T.ClassifierHead(2, C)
        else:
            m = T.TABGNN(channels=C, num_layers=cfg["L"], deg=deg, node_dim=cfg["n_node_cols"] * C, nhidden=C,
                         edge_dim=cfg["n_edge_cols"] * C, nhead=cfg["H"])
            h = T.NodeClassificationHead(cfg["n_classes"], C)
        sd = m.state_dict()
        assert set(sd) == set(cfg["keys"])
        assert all(list(sd[k].shape) == cfg["keys"][k] for k in sd)
        assert set(h.state_dict()) == set(cfg["head_keys"])


def test_constructor_error_behaviour_matches_reference():
    import tabgnn_amd as T
    with pytest.raises(ValueError, match="num_layers must be a positive integer"):     # fused.py:66-68
        T.TABGNNFused(channels=32, num_layers=0, deg=torch.tensor([1, 1]), node_dim=32, nhidden=32, edge_dim=160)
    with pytest.raises(ValueError, match="In degrees are not provided"):                 # utils.py:380-381
        from tabgnn_amd import synthetic as S
        cfg = S.make_config(32, 1, 8, 16)
        cfg["in_degrees"] = None
        T.TABGNNFusedS(cfg)


def test_encoder_launch_plan_and_state_dict_names():
    import tabgnn_amd as T
    from tabgnn_amd import synthetic as S
    enc = T.StypeWiseFeatureEncoder(128, S.EDGE_STATS, S.EDGE_COLS)
    keys = set(enc.state_dict())
    assert {"encoder_dict.numerical.weight", "encoder_dict.numerical.mean", "encoder_dict.categorical.embs.0.weight",
            "encoder_dict.categorical.embs.2.weight", "encoder_dict.timestamp.weight",
            "encoder_dict.timestamp.min_year"} <= keys
    plan = enc._plan
    assert plan["ncols"] == 5 and len(plan["descs"]) == 1
    kinds = [plan["descs"][0].col[i].kind for i in range(5)]
    assert kinds == [0, 1, 1, 1, 2]                                   # numerical, categorical x3, timestamp
    assert plan["acc_floats"][0] == (2 + 16 + 8 + 16 + 57) * 128
    # a wide table (BASELINE config 5: 64 mixed columns, C=256) is split over several launches
    st = T.stype
    names = {st.numerical: [f"n{i}" for i in range(24)], st.categorical: [f"c{i}" for i in range(32)],
             st.timestamp: [f"t{i}" for i in range(8)]}
    stats = {**{f"n{i}": dict(mean=0., std=1.) for i in range(24)},
             **{f"c{i}": dict(cardinality=2 + 37 * i) for i in range(32)},
             **{f"t{i}": dict(min_year=2000) for i in range(8)}}
    wide = T.StypeWiseFeatureEncoder(256, stats, names)
    assert wide._plan["ncols"] == 64 and len(wide._plan["descs"]) >= 4
    assert all(a * 4 <= 150 * 1024 for a in wide._plan["acc_floats"])


def test_synthetic_batch_contract():
    """Seed edges first, every node id present, int64 indices (ibm_transactions_for_aml.py:159-180)."""
    from tabgnn_amd import synthetic as S
    node_tf, ei, edge_tf, y = S.make_batch(200, seed=3)
    assert ei.dtype == torch.int64 and ei.shape == (2, 10702) and node_tf.num_rows == 12797
    assert torch.unique(ei).numel() == node_tf.num_rows and y.shape == (200,)
    assert edge_tf.num_rows == ei.shape[1] and edge_tf.num_cols == 5
    sub = edge_tf[:200, :]
    assert sub.num_rows == 200


_ASM_CACHE = {}


def _gfx950_assembly(name):
    """csrc/<name> compiled to gfx950 assembly with the library's flags (once per session): list of stripped lines."""
    import subprocess
    import tempfile
    if name not in _ASM_CACHE:
        csrc = os.path.join(ROOT, "models-for-relational-multimodal-data_amd", "csrc")
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "k.s")
            subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fno-strict-aliasing", "--offload-arch=gfx950",
                                   "--cuda-device-only", "-S", os.path.join(csrc, name), "-o", out], stderr=subprocess.DEVNULL)
            _ASM_CACHE[name] = [ln.strip() for ln in open(out)]
    return _ASM_CACHE[name]


def test_only_the_dma_helpers_touch_m0():
    """The LDS-DMA helpers (encoder_fused.hip:ef_dma, post_scaled.hip:PS_M0_SET) write M0 from inline assembly, and the
    compiler does not honour an M0 clobber (M0 is a reserved register).  That is only sound while hipcc itself never
    uses M0 in those translation units: compile them to gfx950 assembly and check that every M0 reference is one of the
    helpers' own moves between M0 and an SGPR, and that no instruction that reads M0 IMPLICITLY (relative moves, GWS,
    messages, add-TID LDS forms, the compiler's own LDS-DMA) was generated."""
    import re
    for name in ("post_scaled.hip", "encoder_fused.hip"):
        code = [ln for ln in _gfx950_assembly(name) if ln and not ln.startswith(";")]
        lines = [ln for ln in code if re.search(r"\bm0\b", ln)]
        assert lines, name                                                    # the helpers are there
        other = [ln for ln in lines if not re.fullmatch(r"s_mov_b32 (m0, (s\d+|vcc_lo|vcc_hi)|(s\d+|vcc_lo|vcc_hi), m0)", ln)]
        assert not other, (name, other[:5])
        implicit = [ln for ln in code if re.match(r"(s_movrel|v_movrel|ds_gws|s_sendmsg|s_ttracedata|ds_\w*addtid|v_interp|buffer_load\w* .*\blds\b)", ln)]
        assert not implicit, (name, implicit[:5])


def test_every_barrier_of_the_lds_dma_kernels_waits_for_the_waves_own_lds_operations():
    """Round 5 (LDS-DMA hazard): a raw ``s_barrier`` does not wait for the wave's outstanding LDS reads, and an
    asm ``s_waitcnt vmcnt(K)`` does not stop hipcc from leaving the last ``ds_read_b128`` of a weight unit in flight across
    the barrier that hands the unit's buffer to the next LDS-DMA — which may then land first.  In the two translation units
    that stream operands by LDS-DMA no ``s_barrier`` may have an LDS operation of its own basic block in front of it
    without a wait that includes ``lgkmcnt(0)`` in between (all LDS operations of these files are compiler-visible, so
    across blocks the compiler's waits hold; the raw barriers of the unit / stage boundaries carry theirs in-block)."""
    import re
    for name in ("post_scaled.hip", "encoder_fused.hip"):
        code = [ln for ln in _gfx950_assembly(name) if ln and not ln.startswith(";")]
        # kernels = [function label .. s_endpgm]; only those that issue LDS-DMA themselves are held to the rule (the
        # others synchronise through __syncthreads(), whose waits the compiler places)
        starts = [i for i, ln in enumerate(code) if re.match(r"_Z\w+:", ln)]
        bad, nbar, nker = [], 0, 0
        for a, b in zip(starts, starts[1:] + [len(code)]):
            body = code[a:b]
            if not any(ln.startswith("global_load_lds") for ln in body):
                continue
            nker += 1
            for i, ln in enumerate(body):
                if not ln.startswith("s_barrier"):
                    continue
                nbar += 1
                ok = False
                for j in range(i - 1, -1, -1):
                    t = body[j]
                    if t.startswith("s_waitcnt") and "lgkmcnt(0)" in t:
                        ok = True
                        break
                    if re.match(r"[.\w$]+:", t) or t.startswith(("s_cbranch", "s_branch", "s_barrier")):
                        ok = True                  # block start: nothing of this block is outstanding (the compiler's own
                        break                      # dataflow covers its predecessors; the raw barriers carry their wait in-block)
                    if t.startswith("ds_"):
                        break                      # an LDS operation with no full wait between it and the barrier
                if not ok:
                    bad.append((body[0], body[max(i - 4, 0):i + 1]))
        assert nker >= 2 and nbar >= 8 * nker // 2, (name, nker, nbar)
        assert not bad, (name, len(bad), bad[:3])


def test_no_wide_buffer_store_with_a_register_soffset():
    """Round 5 (store-data hazard): hipcc pads "128-bit store -> write of its data registers" only for stores
    without a register soffset; with one, the VALU instruction behind `buffer_store_dwordx4 v[44:47], ..., s94 offen`
    replaced the store's third dword on loaded CUs (the wrong z1 / z2 rows of rounds 3-4).  The translation units that
    call the raw buffer-store builtin must not produce such a store at all."""
    import re
    csrc = os.path.join(ROOT, "models-for-relational-multimodal-data_amd", "csrc")
    users = [f for f in sorted(os.listdir(csrc)) if f.endswith(".hip") and "raw_buffer_store_b" in open(os.path.join(csrc, f)).read()]
    assert "encoder_fused.hip" in users
    for name in users:
        code = [ln for ln in _gfx950_assembly(name) if ln and not ln.startswith(";")]
        wide = [ln for ln in code if re.match(r"buffer_store_dwordx[34] ", ln)]
        assert len(wide) >= 8, (name, len(wide))
        reg = [ln for ln in wide if re.match(r"buffer_store_dwordx[34] v\[\d+:\d+\], \w+, s\[\d+:\d+\], (s\d+|vcc_lo|vcc_hi|m0|ttmp\d+)\b", ln)]
        assert not reg, (name, len(reg), reg[:3])
