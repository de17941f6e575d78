"""SURVEY 8(f) rows on the GPU: batched on-device mel front end inside the data path, the Synthesizer drop-in and
the TensorBoard stream of a real (tiny) training run."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TEXTS = ["감정있는 한국어 목소리 생성", "안녕하세요 반갑습니다", "오늘 날씨가 좋네요", "테스트 문장입니다", "가나다라마바사", "한국어"]


def _write_wav(path, n, seed):
    from scipy.io.wavfile import write
    g = np.random.RandomState(seed)
    write(path, 16000, (np.clip(0.1 * g.randn(n), -1, 1) * 32767).astype(np.int16))


def _filelist(tmp_path, sizes):
    wavs = []
    for i, n in enumerate(sizes):
        p = str(tmp_path / ('u%d.wav' % i))
        _write_wav(p, n, i)
        wavs.append(p)
    fl = tmp_path / 'koemo_spk_emo_all_test.txt'
    fl.write_text("\n".join("%s|%s|0|%d" % (w, TEXTS[i % len(TEXTS)], i % 4) for i, w in enumerate(wavs)) + "\n",
                  encoding='utf-8')
    return str(fl), wavs


def test_device_frontend_collate_equals_per_utterance_path(tmp_path):
    import hparams as HP
    from data_utils import DeviceFrontendCollate, TextMelCollate, TextMelLoader
    fl, _ = _filelist(tmp_path, (48000, 32011, 40000, 36000, 25600, 600))
    hp = HP.create_hparams()
    host = TextMelLoader(fl, hp)
    dev = TextMelLoader(fl, hp, return_audio=True)
    assert dev.lengths() == [host[i][1].size(1) for i in range(len(host))]          # header-only frame counts
    items_h = [host[i] for i in range(len(host))]
    items_d = [dev[i] for i in range(len(dev))]
    assert items_d[0][1].dtype == torch.int16 and items_d[0][1].dim() == 1
    ref = TextMelCollate(1)(items_h)
    out = DeviceFrontendCollate(hp, stft=dev.stft)(items_d)
    assert out[2].is_cuda and out[3].is_cuda and out[4].is_cuda
    for a, b in zip(ref, out):
        assert a.shape == b.shape and a.dtype == b.dtype
    assert torch.equal(ref[0], out[0]) and torch.equal(ref[1], out[1]) and torch.equal(ref[4], out[4].cpu())
    assert torch.equal(ref[3], out[3].cpu()) and torch.equal(ref[5], out[5]) and torch.equal(ref[6], out[6])
    # int16 * (1/32768) in-kernel vs float division on the host: same fp32 values, so the mels agree to rounding
    assert (ref[2] - out[2].cpu()).abs().max().item() < 2e-5
    with pytest.raises(ValueError):       # shorter than the reflect padding: rejected, like torch's F.pad in the reference
        dev.stft.mel_spectrogram(torch.zeros(1, 400, dtype=torch.int16).cuda(), lengths=torch.tensor([400]), scale=1.0)
    # frames past an utterance are exactly the collate pad value
    for row in range(out[2].size(0)):
        assert float(out[2][row, :, int(out[4][row]):].abs().sum()) == 0.0


def test_train_with_device_frontend_buckets_and_tensorboard(tmp_path, capsys):
    import hparams as HP
    import logger as L
    import train as TR
    fl, _ = _filelist(tmp_path, (48000, 32000, 40000, 36000, 30000, 28000))
    out = str(tmp_path / 'out')
    hp = HP.create_hparams("batch_size=2,anneal_function=constant,epochs=1,iters_per_checkpoint=2,device_frontend=True,"
                           "bucket_batches=True,training_files=%s,validation_files=%s" % (fl, fl))
    TR.train(out, 'logs', None, False, 1, 0, 'group_name', hp)
    printed = capsys.readouterr().out
    assert "Train loss 0 " in printed and "Train loss 2 " in printed and "Validation loss 2:" in printed
    logdir = os.path.join(out, 'logs')
    ev = L.read_events(os.path.join(logdir, os.listdir(logdir)[0]))
    tags = {t for _, t, _, _ in ev}
    assert {"training.loss", "grad.norm", "learning.rate", "duration", "kl_div", "kl_weight", "recon_loss",
            "validation.loss"} <= tags
    assert "decoder/attention_rnn/weight_ih" in tags                     # parameter histograms (logger.py:31-33)
    if L._plots() is not None:
        assert {"alignment", "mel_target", "mel_predicted", "gate", "latent_dim"} <= tags
    losses = [v for s, t, k, v in ev if t == "training.loss"]
    assert len(losses) == 3 and all(np.isfinite(losses))
    assert os.path.isfile(os.path.join(out, 'checkpoint_2'))


def test_synthesizer_drop_in(tmp_path):
    import hparams as HP
    import train as TR
    from synthesizer import Synthesizer
    fl, wavs = _filelist(tmp_path, (30000, 28000, 26000, 24000, 22000, 20000, 18000, 16000))
    hp = HP.create_hparams("max_decoder_steps=40")
    torch.manual_seed(hp.seed)
    model = TR.load_model(hp)
    ck = str(tmp_path / 'ckpt_1000')
    torch.save({'iteration': 1000, 'state_dict': {k: v.detach().clone() for k, v in model.state_dict().items()},
                'optimizer': {}, 'learning_rate': 1e-3}, ck)
    syn = Synthesizer(hp).load(ck, filelist_path=fl)
    cache = Synthesizer.centroid_cache_path(ck, fl)
    assert cache.endswith('ckpt_1000_test.npz') and os.path.isfile(cache)              # reference naming rule
    d = np.load(cache)
    assert d['zs'].shape == (8, hp.z_latent_dim) and d['emotions'].tolist() == [0, 1, 2, 3, 0, 1, 2, 3]
    assert np.allclose(syn.ang, d['zs'][d['emotions'] == 2].mean(0)) and np.allclose(syn.hap, d['zs'][d['emotions'] == 3].mean(0))
    # second load reads the cache (the files are gone)
    for w in wavs[2:]:
        os.remove(w)
    syn2 = Synthesizer(hp).load(ck, filelist_path=fl)
    assert np.array_equal(syn2.neu, syn.neu)
    # emotion-ratio conditioning (ratio order neu, sad, hap, ang) and reference-audio conditioning
    out_path = str(tmp_path / 'gen')
    mel, align = syn.synthesize("안녕하세요", out_path, False, None, (0.5, 0.0, 0.5, 0.0))
    T = mel.size(2)
    assert mel.shape[:2] == (1, 80) and 1 <= T <= 40 and align.shape[0] == 1 and align.shape[1] == T
    assert torch.isfinite(mel).all() and abs(float(align[0, 0].sum()) - 1.0) < 1e-4
    assert np.load(out_path + '.npy').shape == (80, T)
    mel2, _ = syn.synthesize("안녕하세요", None, True, wavs[0], None)
    assert mel2.shape[:2] == (1, 80) and torch.isfinite(mel2).all()
    # a pluggable vocoder gets the pre-Postnet mel, like the reference's WaveGlow call (synthesizer.py:163)
    got = {}
    def vocoder(m):
        got['shape'] = tuple(m.shape)
        return torch.zeros(1, 256 * m.size(2))
    syn.vocoder = vocoder
    wav_path = str(tmp_path / 'gen.wav')
    mel3, _ = syn.synthesize("한국어", wav_path, False, None, (1.0, 0.0, 0.0, 0.0))
    from scipy.io.wavfile import read
    sr, data = read(wav_path)
    assert sr == 16000 and got['shape'] == (1, 80, mel3.size(2)) and len(data) == 256 * mel3.size(2)


@pytest.mark.parametrize("case", ['ratios', 'ref_audio'])
def test_synthesize_matches_the_reference_call_sequence(tmp_path, golden_dir, case):
    """SURVEY 8f-2 end to end: `Synthesizer.synthesize()` against the statements of the reference's
    own `synthesize` (synthesizer.py:112-160) executed by the REAL reference on CPU (tests/golden/synthesize.npz, written by
    oracle/gen_golden.py (h)): emotion-ratio mix of the centroids through fc3 — the loop runs into max_decoder_steps — and
    reference-audio conditioning — the reference's own stop rule ends the run."""
    import hparams as HP
    import model as M
    import train as TR
    from scipy.io.wavfile import write
    from synthesizer import Synthesizer
    g = np.load(os.path.join(golden_dir, 'synthesize.npz'))
    text = bytes(g['text_utf8']).decode('utf-8')
    hp = HP.create_hparams("max_decoder_steps=24")
    old = M.drop_rate
    M.drop_rate = 0.0
    try:
        torch.manual_seed(hp.seed)
        model = TR.load_model(hp)
        ck = str(tmp_path / 'ckpt_5')
        torch.save({'iteration': 5, 'state_dict': {k: v.detach().clone() for k, v in model.state_dict().items()},
                    'optimizer': {}, 'learning_rate': 1e-3}, ck)
        fl = str(tmp_path / 'refs_test.txt')
        np.savez(Synthesizer.centroid_cache_path(ck, fl), zs=g['zs'], emotions=g['emotions'])      # the reference's cache file
        syn = Synthesizer(hp).load(ck, filelist_path=fl)
        em = g['emotions']
        assert np.allclose(syn.neu, g['zs'][em == 0].mean(0)) and np.allclose(syn.hap, g['zs'][em == 3].mean(0))
        with torch.no_grad():
            syn.model.decoder.gate_layer.bias.fill_(float(g[case + '_gate_bias'][0]))
        ref = str(tmp_path / 'ref.wav')
        write(ref, 16000, g['ref_wav'])
        post, align = syn.synthesize(text, None, case == 'ref_audio', ref, tuple(float(r) for r in g['ratios']))
        want_post, want_align = torch.from_numpy(g[case + '_post']), torch.from_numpy(g[case + '_align'])
        assert post.shape == want_post.shape and align.shape == want_align.shape, (post.shape, want_post.shape)
        assert (post.cpu() - want_post).abs().mean() < 1e-4 and (post.cpu() - want_post).abs().max() < 5e-4
        assert (align.cpu() - want_align).abs().max() < 2e-5
    finally:
        M.drop_rate = old


def test_bucket_order_on_the_real_model():
    """1-rank RCCL group on the real model: the Postnet slice of the gradient arena is issued from a hook while
    backward is still running, before the decoder slice, before the encoder slice."""
    import socket
    import sys
    import torch.distributed as dist
    import distributed as D
    import hparams as HP
    import train as TR
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import synthetic_batch
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, world_size=1, rank=0)
    try:
        hp = HP.create_hparams("batch_size=2,anneal_function=constant")
        torch.manual_seed(hp.seed)
        eng = TR.TrainEngine(hp)
        named, offs = eng.optimizer.arena_layout()
        ar = D.OverlappedArenaAllReduce(named, offs, eng.optimizer.grads, force=True,
                                        side_streams=lambda: [st for st in (getattr(eng.model, '_side', None),) if st],
                                        gather=eng.optimizer.gather_grads)
        eng.allreduce = ar
        batch = synthetic_batch(2, 12, 30, 3)
        opt = eng.optimizer
        opt.zero_grad()
        x, y = eng.model.parse_batch(batch)
        loss = eng.criterion(eng.model(x), y, 0)[0]
        ar.begin()
        loss.backward()
        ar.finish()
        opt.mark_gathered()
        ref = opt.grads.clone()
        # the bucket-by-bucket gather filled the arena with exactly autograd's gradients
        for (n, p), o in zip(named, offs):
            assert torch.equal(ref[o:o + p.numel()].view_as(p), p.grad), n
        opt.step()
        torch.cuda.synchronize()
        names = [b[0] for b in ar.buckets]
        assert names == ['transcript_embedding+encoder', 'decoder', 'postnet', 'vae_gst']
        assert ar.buckets[0][1] == 0 and ar.buckets[-1][2] == eng.optimizer.grads.numel()
        order = [names[bi] for bi, _ in ar.launch_log]
        assert order.index('postnet') < order.index('decoder') < order.index('transcript_embedding+encoder')
        assert all(from_hook for _, from_hook in ar.launch_log)
        assert torch.isfinite(ref).all()
    finally:
        dist.destroy_process_group()
