"""Host-side mirror of the reference interface: hparams, collate layout, loss schedule, C-ABI exports,
product path refuses to run without its HIP backend.  CPU only, no compute through the library."""
import json
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_hparams_defaults_and_override(golden_dir):
    import hparams as HP
    with open(os.path.join(golden_dir, 'hparams_defaults.json')) as f:
        g = json.load(f)
    assert HP.create_hparams().values() == g['defaults']
    assert HP.create_hparams(g['override_string']).values() == g['overridden']
    with pytest.raises(ValueError):
        HP.create_hparams("no_such_param=1")
    hp = HP.create_hparams("distributed_run=True,fp16_run=0")
    assert hp.distributed_run is True and hp.fp16_run is False


def test_collate_layout(golden_dir):
    from data_utils import TextMelCollate
    g = np.load(os.path.join(golden_dir, 'collate.npz'))
    items = [(torch.from_numpy(g['in_text_%d' % i]), torch.from_numpy(g['in_mel_%d' % i]), torch.tensor([1.0]),
              torch.from_numpy(g['in_emo_%d' % i])) for i in range(3)]
    out = TextMelCollate(1)(items)
    assert len(out) == 7
    for i, t in enumerate(out):
        assert str(t.dtype) == str(g['out_dtypes'][i]), i
        assert np.array_equal(t.numpy(), g['out_%d' % i]), i
    # n_frames_per_step rounding (data_utils.py:120-122)
    assert TextMelCollate(4)(items)[2].shape[2] % 4 == 0


def test_kl_anneal_schedule(golden_dir):
    import hparams as HP
    from loss_function import Tacotron2Loss_VAE
    with open(os.path.join(golden_dir, 'kl_anneal.json')) as f:
        g = json.load(f)
    hp = HP.create_hparams()
    crit = Tacotron2Loss_VAE(hp)
    steps = (0, 1, 5000, 10000, 20000, 50000, 50001, 100000)
    for kind, vals in g.items():
        got = [crit.kl_anneal_function(kind, hp.anneal_lag, s, hp.anneal_k, hp.anneal_x0, hp.anneal_upper) for s in steps]
        assert got == vals, kind


def test_loss_matches_oracle_on_cpu():
    """The loss module is plain host-side torch; compare with the oracle restatement."""
    import hparams as HP
    import t2v_oracle as O
    from loss_function import Tacotron2Loss_VAE
    g = torch.Generator().manual_seed(0)
    B, T = 3, 17
    outs = [torch.randn(B, 80, T, generator=g), torch.randn(B, 80, T, generator=g), torch.randn(B, T, generator=g),
            None, torch.randn(B, 32, generator=g), torch.randn(B, 32, generator=g) * 0.1, None, None]
    tgt = (torch.randn(B, 80, T, generator=g), (torch.rand(B, T, generator=g) > 0.8).float())
    crit = Tacotron2Loss_VAE(HP.create_hparams("anneal_function=logistic"))
    a = crit(outs, tgt, 12345)
    b = O.loss_forward(outs, tgt[0], tgt[1], 12345, 'logistic')
    for x, y in zip(a, b):
        assert float(x) == pytest.approx(float(y), rel=1e-6)


def test_c_abi_exports_every_declared_symbol():
    """include/t2vae.h is the contract: every function it declares must be exported by the in-tree
    library (dlopen only — no kernel is launched here)."""
    import t2v_hip
    with open(os.path.join(ROOT, 'include', 't2vae.h')) as f:
        src = re.sub(r'/\*.*?\*/', '', f.read(), flags=re.S)
    declared = set(re.findall(r'\b(t2v_[a-z0-9_]+)\s*\(', src))
    assert len(declared) >= 7
    lib = t2v_hip.load_library()
    for name in declared:
        assert hasattr(lib, name), name
    assert b'gfx950' in lib.t2v_version()


def test_ctypes_bindings_match_header_arity():
    """Every prototype in include/t2vae.h takes as many parameters as the ctypes binding passes (ABI drift guard:
    ctypes would silently push a wrong argument list)."""
    import t2v_hip
    with open(os.path.join(ROOT, 'include', 't2vae.h')) as f:
        src = re.sub(r'/\*.*?\*/', '', f.read(), flags=re.S)
    protos = dict(re.findall(r'\b(t2v_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;', src, flags=re.S))
    lib = t2v_hip.load_library()
    checked = 0
    for name, params in protos.items():
        fn = getattr(lib, name)
        if fn.argtypes is None:
            continue
        params = params.strip()
        n = 0 if params in ('', 'void') else params.count(',') + 1
        assert n == len(fn.argtypes), (name, n, len(fn.argtypes))
        checked += 1
    assert checked >= 20


def test_product_path_has_no_cpu_fallback():
    import hparams as HP
    import model as M
    import t2v_hip
    hp = HP.create_hparams()
    torch.manual_seed(0)
    dec = M.Decoder(hp)
    with pytest.raises(t2v_hip.T2VHipError):
        dec(torch.zeros(1, 5, 512), torch.zeros(1, 80, 3), torch.tensor([5]))
    if not torch.cuda.is_available():
        import train as TR
        with pytest.raises(RuntimeError):
            TR.load_model(hp)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'tacotron2-vae_amd')
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith(('.py', '.hip', '.h')):
                with open(os.path.join(dirpath, fn), encoding='utf-8') as f:
                    src = f.read()
                assert 't2v_oracle' not in src and 'ref_shims' not in src and '_refimport' not in src, fn


def _header_structs():
    """{struct name: [(field name, is_pointer)]} parsed from include/t2vae.h"""
    import re
    text = open(os.path.join(ROOT, 'include', 't2vae.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    out = {}
    for m in re.finditer(r'typedef struct (\w+) \{(.*?)\} \1;', text, flags=re.S):
        fields = []
        for decl in m.group(2).split(';'):
            decl = decl.strip()
            if not decl:
                continue
            name = re.search(r'(\w+)\s*(\[\d+\])?$', decl).group(1)
            fields.append((name, '*' in decl))
        out[m.group(1)] = fields
    return out


def test_ctypes_structs_match_header_in_package_and_integration_stub():
    """A ctypes.Structure that is one field short passes garbage for the missing member (the INTEGRATION stub
    lacked t2v_dec_weights.packs_bf16).  Every Structure of t2v_hip.py and every `class X(C.Structure):  # t2v_...` of
    INTEGRATION.md must list the header's fields, in order, pointers as c_void_p."""
    import ctypes as C
    import re
    import t2v_hip
    hdr = _header_structs()
    pairs = {'_DecWeights': 't2v_dec_weights', '_DecTrainBufs': 't2v_dec_train_bufs', '_DecBwdBufs': 't2v_dec_bwd_bufs',
             '_DecPersistWeights': 't2v_dec_persist_weights', '_DecPersistBufs': 't2v_dec_persist_bufs',
             '_DecInferBufs': 't2v_dec_infer_bufs', '_DecTrainPersistWeights': 't2v_dec_train_persist_weights'}
    for cls_name in [n for n in dir(t2v_hip) if isinstance(getattr(t2v_hip, n), type) and issubclass(getattr(t2v_hip, n), C.Structure)
                     and n != 'Structure']:
        assert (cls_name in pairs or cls_name.lstrip('_') in [k.lstrip('_') for k in pairs] or cls_name.startswith('_Dec')
                or hasattr(getattr(t2v_hip, cls_name), 'C_NAME')), cls_name
    named = [n for n in dir(t2v_hip) if isinstance(getattr(t2v_hip, n), type) and issubclass(getattr(t2v_hip, n), C.Structure)
             and hasattr(getattr(t2v_hip, n), 'C_NAME') and n not in pairs]
    is_ptr = lambda t: t is C.c_void_p or getattr(t, '_type_', None) is C.c_void_p      # (a pointer, or an array of pointers)
    for cls_name, sname in list(pairs.items()) + [(n, None) for n in named]:
        cls = getattr(t2v_hip, cls_name)
        if sname is None:
            sname = getattr(cls, 'C_NAME')
        want = hdr[sname]
        got = [(n, is_ptr(t)) for n, t in cls._fields_]
        assert got == want, (cls_name, got, want)
    # the stub a maintainer copies out of INTEGRATION.md
    doc = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    stubs = re.findall(r'class (\w+)\(C\.Structure\):\s*#\s*(t2v_\w+)\n(.*?)(?=\nclass |\n\n)', doc, flags=re.S)
    assert len(stubs) >= 2
    for cls_name, sname, body in stubs:
        ns = {'C': C}
        exec('class %s(C.Structure):\n%s' % (cls_name, body), ns)
        got = [(n, is_ptr(t)) for n, t in ns[cls_name]._fields_]
        assert got == hdr[sname], (cls_name, got, hdr[sname])
        mine = {v: k for k, v in pairs.items()}.get(sname) or next(n for n in named if getattr(t2v_hip, n).C_NAME == sname)
        assert C.sizeof(ns[cls_name]) == C.sizeof(getattr(t2v_hip, mine))


def test_limit_host_threads_only_lowers_and_honours_the_env(monkeypatch):
    """The training engine / inference session cap torch's intra-op threads (128 spinning OpenMP workers exhaust
    the CPU quota of the GPU boxes' containers and freeze the process every 100 ms).  The cap only ever LOWERS the count,
    T2V_HOST_THREADS overrides it, 0 leaves torch alone, a multi-rank job defaults to 2 per rank."""
    import torch
    import t2v_hip
    old = torch.get_num_threads()
    try:
        torch.set_num_threads(6)
        monkeypatch.delenv('T2V_HOST_THREADS', raising=False)
        monkeypatch.setenv('WORLD_SIZE', '1')
        t2v_hip.limit_host_threads()
        assert torch.get_num_threads() == 4
        t2v_hip.limit_host_threads(8)                 # never raises the count
        assert torch.get_num_threads() == 4
        monkeypatch.setenv('WORLD_SIZE', '8')
        t2v_hip.limit_host_threads()
        assert torch.get_num_threads() == 2
        torch.set_num_threads(6)
        monkeypatch.setenv('T2V_HOST_THREADS', '0')
        t2v_hip.limit_host_threads()
        assert torch.get_num_threads() == 6
        monkeypatch.setenv('T2V_HOST_THREADS', '3')
        t2v_hip.limit_host_threads()
        assert torch.get_num_threads() == 3
    finally:
        torch.set_num_threads(old)


def test_batch_layout_packs_the_seven_tensors_and_int32_lengths():
    """model.BatchLayout (the one-buffer form of reference model.py:486-503 parse_batch): every field 16-byte aligned, dtypes
    converted on the host, views read back what was packed, and both length vectors additionally as int32 (round 4)"""
    import model as M
    g = torch.Generator().manual_seed(0)
    text = torch.randint(0, 70, (3, 11), generator=g)
    in_len = torch.tensor([11, 9, 4])
    mel = torch.randn(3, 80, 17, generator=g)
    gate = torch.rand(3, 17, generator=g)
    out_len = torch.tensor([17, 12, 5])
    spk = torch.zeros(3, 4, dtype=torch.long); spk[:, 1] = 1
    emo = torch.zeros(3, 5, dtype=torch.long); emo[:, 2] = 1
    batch = (text, in_len, mel, gate, out_len, spk, emo)
    lay = M.BatchLayout(batch)
    assert lay.nbytes % 16 == 0 and all(off % 16 == 0 for off, _, _, _ in lay.fields) and all(off % 16 == 0 for _, off, _ in lay.i32)
    host = torch.zeros(lay.nbytes, dtype=torch.uint8)
    lay.pack_into(host, batch)
    (t2, l2, m2, max_len, o2, s2, e2), (m3, g3) = lay.views(host)
    assert max_len == 11 and torch.equal(t2, text) and torch.equal(l2, in_len) and torch.equal(o2, out_len)
    assert torch.equal(m2, mel) and torch.equal(g3, gate) and m3 is m2
    assert s2.dtype == torch.float32 and torch.equal(s2, spk.float()) and torch.equal(e2, emo.float())
    assert l2._i32.dtype == torch.int32 and l2._i32.tolist() == [11, 9, 4] and o2._i32.tolist() == [17, 12, 5]
    assert M.lengths_i32(l2, torch.device('cpu')) is l2._i32
    plain = torch.tensor([3, 2])
    assert M.lengths_i32(plain, torch.device('cpu')).dtype == torch.int32         # no companion: converted
    assert lay.key == M.BatchLayout(batch).key


def test_error_ledger_blocks_are_recycled_and_slots_stay_in_bounds():
    """captured graphs used to pin ledger slots for the life of the process while an LRU evicted the
    graphs themselves — a ragged loader walked the write index past the 4096-word ledger.  Now a graph owns a fixed block
    that goes back to a free list when the graph is evicted, eager notes live in their own region, a step reserves a
    contiguous run (the ledger is checked and restarted before it would wrap), and a capture without a free block is refused.
    Pure host logic: exercised here on a CPU-resident ledger."""
    import t2v_hip as H
    dev = torch.device('cpu')
    H._ERR_POOL.pop('cpu', None)
    _, pool = H._err_pool(dev)
    n_blocks = H._ERR_GRAPH_SLOTS // H._ERR_BLOCK
    # captures: each takes a block; its notes land inside; the span is what the engine reads after a replay
    spans = []
    for g in range(n_blocks):
        assert H.err_capture_begin(dev)
        m = H.err_mark(dev)
        for k in range(7):
            s = H._err_take_slot(pool)
            pool.labels[s] = 'graph %d note %d' % (g, k)
            assert m <= s < m + H._ERR_BLOCK <= H._ERR_GRAPH_SLOTS
        assert H.err_range(m, dev).numel() == 7
        spans.append(H.err_capture_end(dev))
        assert spans[-1] == (m, m + 7)
    assert len({s[0] for s in spans}) == n_blocks
    assert not H.err_capture_begin(dev)                 # no block left: the engine keeps that shape eager
    # an evicted graph frees its block (labels gone, words cleared), the next capture gets it
    pool.words[spans[3][0]] = 1
    H.err_release(spans[3], dev)
    assert spans[3][0] not in pool.labels and int(pool.words[spans[3][0]]) == 0
    assert H.err_capture_begin(dev)
    assert H.err_mark(dev) == spans[3][0]
    # a capture that outgrows its block is an error, not a silent walk into the neighbour's words
    for _ in range(H._ERR_BLOCK):
        H._err_take_slot(pool)
    with pytest.raises(H.T2VHipError):
        H._err_take_slot(pool)
    assert H.err_capture_end(dev, keep=False) is None
    assert spans[3][0] in pool.free_blocks
    # eager notes: never below the graph region, never at or past the end; a step's run is contiguous
    seen = []
    for step in range(3000):
        m = H.err_mark(dev)
        assert H._ERR_GRAPH_SLOTS <= m and m + H._ERR_STEP_RESERVE <= H._ERR_SLOTS
        for _ in range(7):
            s = H._err_take_slot(pool)
            assert m <= s < H._ERR_SLOTS
            seen.append(s)
        r = H.err_range(m, dev)
        assert r is not None and r.numel() == 7
    assert min(seen) == H._ERR_GRAPH_SLOTS and max(seen) < H._ERR_SLOTS
    # a set word anywhere (graph block or eager region) is reported by the next check, with its label, and cleared
    pool.words[spans[5][0] + 2] = 1
    with pytest.raises(H.T2VHipError) as ei:
        H.check_async_errors()
    assert 'graph 5 note 2' in str(ei.value)
    H.check_async_errors()
    H._ERR_POOL.pop('cpu', None)
