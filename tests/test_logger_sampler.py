"""Host-side pieces of SURVEY 8(f): the self-contained TensorBoard event writer (tags of reference
logger.py:9-56), the length-bucketed batch sampler and the build-specific hparams switches."""
import os

import numpy as np
import torch


def test_crc32c_known_answers():
    import logger as L
    assert L._crc32c(b"123456789") == 0xE3069283            # RFC 3720 check value
    assert L._crc32c(b"") == 0
    assert L._crc32c(bytes(32)) == 0x8A9136AA               # 32 zero bytes (RFC 3720 B.4)


def test_event_file_round_trip(tmp_path):
    import logger as L
    lg = L.Tacotron2Logger(str(tmp_path / 'logs'))
    lg.log_training(1.5, 0.25, 1e-3, 0.03, 1.4, 12.0, 0.001, 7)
    lg.add_histogram('decoder/x', np.arange(100, dtype=np.float32), 7)
    img = np.zeros((4, 6, 3), dtype=np.uint8)
    img[1, 2] = (255, 0, 0)
    lg.add_image('alignment', img, 7)
    lg.close()
    files = os.listdir(str(tmp_path / 'logs'))
    assert len(files) == 1 and files[0].startswith('events.out.tfevents.')
    ev = L.read_events(os.path.join(str(tmp_path / 'logs'), files[0]))        # verifies every record's CRCs
    scal = {t: v for s, t, k, v in ev if k == 'scalar'}
    # the reference's training tags, in its order (logger.py:15-21)
    assert [t for s, t, k, v in ev if k == 'scalar'] == ["training.loss", "grad.norm", "learning.rate", "duration",
                                                          "kl_div", "kl_weight", "recon_loss"]
    assert abs(scal["training.loss"] - 1.5) < 1e-7 and abs(scal["kl_div"] - 12.0) < 1e-6
    assert abs(scal["recon_loss"] - 1.4) < 1e-6 and abs(scal["learning.rate"] - 1e-3) < 1e-9
    assert all(s == 7 for s, _, _, _ in ev)
    hist = [v for s, t, k, v in ev if k == 'histo'][0]
    assert hist[1] == 0.0 and hist[2] == 99.0 and hist[3] == 100.0 and hist[4] == 4950.0
    h, w, sig = [v for s, t, k, v in ev if k == 'image'][0]
    assert (h, w) == (4, 6) and sig == b'\x89PNG\r\n\x1a\n'


def test_png_encoder_decodes_with_matplotlib(tmp_path):
    import logger as L
    plt = L._plots()
    if plt is None:
        return
    img = (np.random.RandomState(0).rand(9, 13, 3) * 255).astype(np.uint8)
    p = str(tmp_path / 'x.png')
    with open(p, 'wb') as f:
        f.write(L._png(img))
    back = plt.imread(p)
    assert back.shape[:2] == (9, 13) and np.abs(back[:, :, :3] * 255 - img).max() < 0.51
    a = L.plot_alignment_to_numpy(np.random.rand(20, 30))
    assert a.ndim == 3 and a.shape[2] == 3 and a.dtype == np.uint8


def test_bucket_sampler_partitions_every_epoch():
    from data_utils import BucketBatchSampler
    rs = np.random.RandomState(1)
    lengths = rs.randint(100, 900, size=1003).tolist()
    world, bs = 4, 6
    per_rank = [list(BucketBatchSampler(lengths, bs, world_size=world, rank=r, seed=5)) for r in range(world)]
    assert len({len(b) for b in per_rank}) == 1 and len(per_rank[0]) == 1003 // (bs * world)
    seen = [i for rank in per_rank for batch in rank for i in batch]
    assert len(seen) == len(set(seen)) == (1003 // (bs * world)) * bs * world        # disjoint, drop_last
    assert all(len(b) == bs for rank in per_rank for b in rank)
    # bucketing: the spread of lengths inside a global batch is far below the spread of the data set, and the
    # ranks of one step see similar amounts of work
    spreads, imbalance = [], []
    for step in range(len(per_rank[0])):
        gl = [lengths[i] for r in range(world) for i in per_rank[r][step]]
        spreads.append(max(gl) - min(gl))
        work = [sum(lengths[i] for i in per_rank[r][step]) for r in range(world)]
        imbalance.append(max(work) / (sum(work) / world))
    assert np.mean(spreads) < 0.25 * (max(lengths) - min(lengths))
    assert np.mean(imbalance) < 1.05
    # a new epoch reshuffles; the same epoch is reproducible
    s = BucketBatchSampler(lengths, bs, world_size=world, rank=0, seed=5)
    e0 = list(s)
    s.set_epoch(1)
    e1 = list(s)
    s.set_epoch(0)
    assert e0 != e1 and list(s) == e0


def test_hparams_extensions_do_not_leak_into_reference_values():
    import hparams as HP
    hp = HP.create_hparams("device_frontend=True,bucket_batches=1,batch_size=4")
    assert hp.device_frontend is True and hp.bucket_batches is True and hp.bf16_run is False
    assert 'device_frontend' not in hp.values() and hp.values()['batch_size'] == 4
    assert set(hp.extensions()) == {'device_frontend', 'bucket_batches', 'bf16_run', 'graph_step', 'fp32_allreduce'}


def test_bucket_sampler_keeps_long_texts_together():
    """round 4: with text lengths given, utterances above a text cap (the persistent decoder kernels' range: 224 symbols then, 560
    — the default cap — since round 6) share batches instead of being spread over many; every utterance is still drawn exactly
    once per epoch"""
    import random
    from data_utils import BucketBatchSampler
    rnd = random.Random(3)
    n, bs = 960, 6
    lengths = [rnd.randint(100, 800) for _ in range(n)]
    texts = [rnd.randint(230, 555) if rnd.random() < 0.1 else rnd.randint(10, 200) for _ in range(n)]
    plain = BucketBatchSampler(lengths, bs, seed=5)
    aware = BucketBatchSampler(lengths, bs, seed=5, text_lengths=texts, text_cap=224)
    assert BucketBatchSampler(lengths, bs, seed=5, text_lengths=texts).persistent_hit_rate() == 1.0      # default cap 560
    seen = sorted(i for b in aware for i in b)
    assert seen == list(range(n))
    hit_aware = aware.persistent_hit_rate()
    hit_plain = sum(max(texts[i] for i in b) <= 224 for b in plain) / len(plain)
    n_long = sum(t > 224 for t in texts)
    assert plain.persistent_hit_rate() is None
    assert hit_aware > hit_plain + 0.2
    assert hit_aware >= 1.0 - (n_long / bs + n // (bs * 16) + 1) / (n // bs)       # at most one mixed batch per window
