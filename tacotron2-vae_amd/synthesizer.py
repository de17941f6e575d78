"""Inference wrapper with the call sequence of the reference's `Synthesizer` (reference synthesizer.py:74-168,
README.md:157-253): load a checkpoint, build (or read) the per-emotion latent centroids, then synthesise a
mel from text conditioned either on a reference utterance or on an emotion-ratio mix.

Differences by design:
* the decode loop runs in the HIP inference session (`Decoder.inference`, 4 launches per frame) instead of
  Python-stepping `decode()`; the stepwise API is still there for callers that want it;
* the vocoder is pluggable: any callable mel(1,80,T) -> waveform.  WaveGlow is an unpinned submodule of the
  reference and is out of scope here; without a vocoder `synthesize` writes the mel as `<path>.npy`.
"""
import os

import numpy as np
import torch

from hparams import create_hparams
from layers import TacotronSTFT
from text import text_to_sequence
from utils import load_wav_to_torch

EMOTIONS = ('neu', 'sad', 'ang', 'hap')      # label ids 0..3 of the koemo filelists


class Synthesizer(object):
    def __init__(self, hparams=None):
        if hparams is None:             # the reference's constructor (synthesizer.py:47-51): defaults + two overrides
            hparams = create_hparams()
            hparams.sampling_rate = 16000
            hparams.max_decoder_steps = 600
        self.hparams = hparams
        hp = self.hparams
        self.stft = TacotronSTFT(hp.filter_length, hp.hop_length, hp.win_length, hp.n_mel_channels,
                                 hp.sampling_rate, hp.mel_fmin, hp.mel_fmax)
        self.model = None
        self.vocoder = None
        self.neu = self.sad = self.ang = self.hap = None

    # ------------------------------------------------------------------ audio -> mel (synthesizer.py:57-68)
    def load_mel(self, path):
        audio, sampling_rate = load_wav_to_torch(path)
        if sampling_rate != self.hparams.sampling_rate:
            raise ValueError("{} SR doesn't match target {} SR".format(sampling_rate, self.hparams.sampling_rate))
        audio_norm = (audio / self.hparams.max_wav_value).unsqueeze(0)
        return self.stft.mel_spectrogram(audio_norm.cuda())

    # ------------------------------------------------------------------ checkpoint + centroids (synthesizer.py:74-110)
    @staticmethod
    def centroid_cache_path(checkpoint_path, filelist_path):
        """`<dir of ckpt>/<ckpt name>_<last '_' field of the filelist name, sans extension>.npz`"""
        suffix = filelist_path.rsplit('_', 1)[1].split('.')[0] if '_' in filelist_path else 'refs'
        return os.path.join(os.path.dirname(checkpoint_path), os.path.basename(checkpoint_path) + '_' + suffix + '.npz')

    def load(self, checkpoint_path, waveglow_path=None, vocoder=None,
             filelist_path='./web/static/uploads/koemo_spk_emo_all_test.txt'):
        """Positional order of the reference (synthesizer.py:74: `load(checkpoint_path, waveglow_path)`, called from
        app.py:161).  waveglow_path: a WaveGlow checkpoint `{'model': module}` exactly as the reference loads it
        (needs the `waveglow` package importable: it is an un-vendored submodule of the reference); `vocoder`: any
        callable mel (1,80,T) -> audio instead.  A callable passed in the second position is taken as the vocoder."""
        from train import load_model
        self.model = load_model(self.hparams)
        self.model.load_state_dict(torch.load(checkpoint_path, map_location='cpu')['state_dict'])
        self.model.eval()
        if callable(waveglow_path) and vocoder is None:
            vocoder, waveglow_path = waveglow_path, None
        self.waveglow = None
        if waveglow_path is not None:
            try:
                self.waveglow = torch.load(waveglow_path, map_location='cpu', weights_only=False)['model'].cuda()
            except Exception as e:
                raise RuntimeError("cannot load the WaveGlow checkpoint %r (the reference's vocoder is an un-vendored "
                                   "submodule; its package must be importable): %s" % (waveglow_path, e))
            if vocoder is None:
                waveglow = self.waveglow
                vocoder = lambda mel: waveglow.infer(mel, sigma=0.666)      # reference synthesizer.py:163
        self.vocoder = vocoder
        npz_path = self.centroid_cache_path(checkpoint_path, filelist_path)
        if os.path.exists(npz_path):
            d = np.load(npz_path)
            zs, emotions = d['zs'], d['emotions']
        else:
            with open(filelist_path, encoding='utf-8') as f:
                rows = [line.strip().split("|") for line in f if line.strip()]
            zs, emotions = [], []
            with torch.no_grad():
                for audio_path, _, _, emotion in rows:
                    _, _, _, z = self.model.vae_gst(self.load_mel(audio_path))
                    zs.append(z.detach().cpu())
                    emotions.append(int(emotion))
            emotions = np.array(emotions)
            zs = torch.cat(zs, dim=0).numpy()
            np.savez(npz_path, zs=zs, emotions=emotions)
        for i, name in enumerate(EMOTIONS):
            sel = zs[emotions == i, :]
            setattr(self, name, np.mean(sel, axis=0) if len(sel) else np.zeros(zs.shape[1], dtype=zs.dtype))
        return self

    # ------------------------------------------------------------------ text -> mel (synthesizer.py:112-168)
    def encode_text(self, text):
        sequence = np.array(text_to_sequence(text, ['korean_cleaners']))[None, :]
        sequence = torch.from_numpy(sequence).cuda().long()
        inputs = self.model.parse_input(sequence)
        embedded = self.model.transcript_embedding(inputs).transpose(1, 2)
        return self.model.encoder.inference(embedded)

    def style_vector(self, transcript_outputs, condition_on_ref, ref_audio, ratios):
        if condition_on_ref:
            latent, _, _, _ = self.model.vae_gst(self.load_mel(ref_audio))
            return latent.unsqueeze(1).expand_as(transcript_outputs)
        # ratio order of the reference: (neu, sad, hap, ang) — synthesizer.py:129-130
        mix = ratios[0] * self.neu + ratios[1] * self.sad + ratios[2] * self.hap + ratios[3] * self.ang
        return self.model.vae_gst.fc3(torch.as_tensor(mix, dtype=torch.float32).cuda())

    @torch.no_grad()
    def synthesize(self, text, path=None, condition_on_ref=False, ref_audio=None, ratios=(1.0, 0.0, 0.0, 0.0)):
        """Returns (mel_outputs_postnet (1,80,T), alignments (1,T,T_in)); writes `path` (wav through the vocoder,
        else `<path>.npy`) when a path is given."""
        transcript_outputs = self.encode_text(text)
        encoder_outputs = transcript_outputs + self.style_vector(transcript_outputs, condition_on_ref, ref_audio, ratios)
        mel_outputs, gate_outputs, alignments = self.model.decoder.inference(encoder_outputs)
        mel_outputs_postnet = mel_outputs + self.model.postnet(mel_outputs)
        if path is not None:
            if self.vocoder is not None:
                from scipy.io.wavfile import write
                audio = self.vocoder(mel_outputs)
                audio = audio[0] if torch.is_tensor(audio) and audio.dim() > 1 else audio
                write(path, self.hparams.sampling_rate, np.asarray(torch.as_tensor(audio).detach().cpu().float()))
            else:
                np.save(path + '.npy', mel_outputs_postnet[0].detach().cpu().numpy())
        return mel_outputs_postnet, alignments
