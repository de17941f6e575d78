"""Dataset + collate with the reference's batch layout (reference data_utils.py:11-137)."""
import random

import numpy as np
import torch
import torch.utils.data

from utils import load_filepaths_and_text, load_wav_to_torch


class TextMelLoader(torch.utils.data.Dataset):
    """filelist line `path|text|speaker|emotion` → (ids IntTensor, mel (80,T), one-hot speaker,
    one-hot emotion).  The list is shuffled once with random.seed(1234) like data_utils.py:29-30."""

    def __init__(self, audiopaths_and_text, hparams, stft=None, return_audio=False):
        self.audiopaths_and_text = load_filepaths_and_text(audiopaths_and_text)
        self.return_audio = bool(return_audio)      # hand raw int16 PCM to DeviceFrontendCollate instead of a mel
        self.text_cleaners = hparams.text_cleaners
        self.max_wav_value = hparams.max_wav_value
        self.sampling_rate = hparams.sampling_rate
        self.load_mel_from_disk = hparams.load_mel_from_disk
        self.n_speakers, self.n_emotions = hparams.n_speakers, hparams.n_emotions
        self.n_mel_channels = hparams.n_mel_channels
        if stft is None and not self.load_mel_from_disk:
            import layers
            stft = layers.TacotronSTFT(hparams.filter_length, hparams.hop_length, hparams.win_length,
                                       hparams.n_mel_channels, hparams.sampling_rate, hparams.mel_fmin,
                                       hparams.mel_fmax)
        self.stft = stft
        random.seed(1234)
        random.shuffle(self.audiopaths_and_text)

    def get_mel(self, filename):
        if self.load_mel_from_disk:
            mel = torch.from_numpy(np.load(filename))
            if mel.size(0) != self.n_mel_channels:
                raise AssertionError('Mel dimension mismatch: given {}, expected {}'.format(
                    mel.size(0), self.n_mel_channels))
            return mel
        audio, sr = load_wav_to_torch(filename)
        if sr != self.sampling_rate:
            raise ValueError("{} SR doesn't match target {} SR".format(sr, self.sampling_rate))
        return self.stft.mel_spectrogram((audio / self.max_wav_value).unsqueeze(0)).squeeze(0)

    def get_audio(self, filename):
        """raw PCM samples as an int16 tensor (the batched device front end scales by 1/max_wav_value itself)"""
        from scipy.io.wavfile import read
        sr, data = read(filename)
        if sr != self.sampling_rate:
            raise ValueError("{} SR doesn't match target {} SR".format(sr, self.sampling_rate))
        if data.dtype != np.int16:
            raise ValueError("{}: the device front end takes 16-bit PCM".format(filename))
        return torch.from_numpy(np.ascontiguousarray(data))

    def lengths(self):
        """output length (mel frames) of every entry without decoding audio: from the wav header / npy header"""
        import wave
        out = []
        for fields in self.audiopaths_and_text:
            if self.load_mel_from_disk:
                out.append(int(np.load(fields[0], mmap_mode='r').shape[1]))
            else:
                with wave.open(fields[0], 'rb') as w:
                    out.append(w.getnframes() // 256 + 1)
        return out

    def text_lengths(self):
        """symbol count of every entry (the decoder's T_in): the bucketed sampler keeps utterances above the persistent
        decoder kernels' range (560 symbols since round 6; 224 before) together, so that as many batches as possible stay inside it"""
        from text import text_to_sequence
        return [len(text_to_sequence(fields[1], self.text_cleaners)) for fields in self.audiopaths_and_text]

    def get_text(self, text):
        from text import text_to_sequence
        return torch.IntTensor(text_to_sequence(text, self.text_cleaners))

    @staticmethod
    def _one_hot(index, n):
        v = torch.zeros(n, dtype=torch.float32)
        v[int(index)] = 1
        return v

    def get_speaker(self, speaker):
        return self._one_hot(speaker, self.n_speakers)

    def get_emotion(self, emotion):
        return self._one_hot(emotion, self.n_emotions)

    def get_mel_text_pair(self, fields):
        path, text, speaker, emotion = fields[0], fields[1], fields[2], fields[3]
        feat = self.get_audio(path) if self.return_audio and not self.load_mel_from_disk else self.get_mel(path)
        return (self.get_text(text), feat, self.get_speaker(speaker), self.get_emotion(emotion))

    def __getitem__(self, index):
        return self.get_mel_text_pair(self.audiopaths_and_text[index])

    def __len__(self):
        return len(self.audiopaths_and_text)


class TextMelCollate(object):
    """Sort by text length (descending), right-pad ids with 0 and mels with 0.0, gate = 1 from the
    last real frame on; returns the reference's 7-tuple with its dtypes (speakers/emotions int64,
    Appendix B-11)."""

    def __init__(self, n_frames_per_step):
        self.n_frames_per_step = n_frames_per_step

    def __call__(self, batch):
        n = len(batch)
        input_lengths, order = torch.sort(torch.LongTensor([len(x[0]) for x in batch]), dim=0, descending=True)
        text_padded = torch.zeros(n, int(input_lengths[0]), dtype=torch.long)
        speakers = torch.zeros(n, len(batch[0][2]), dtype=torch.long)
        emotions = torch.zeros(n, len(batch[0][3]), dtype=torch.long)
        num_mels = batch[0][1].size(0)
        max_t = max(x[1].size(1) for x in batch)
        r = self.n_frames_per_step
        if max_t % r:
            max_t += r - max_t % r
        mel_padded = torch.zeros(n, num_mels, max_t, dtype=torch.float32)
        gate_padded = torch.zeros(n, max_t, dtype=torch.float32)
        output_lengths = torch.zeros(n, dtype=torch.long)
        for row, src in enumerate(order.tolist()):
            text, mel, spk, emo = batch[src]
            text_padded[row, :text.size(0)] = text
            speakers[row] = spk
            emotions[row] = emo
            t = mel.size(1)
            mel_padded[row, :, :t] = mel
            gate_padded[row, t - 1:] = 1
            output_lengths[row] = t
        return text_padded, input_lengths, mel_padded, gate_padded, output_lengths, speakers, emotions


class DeviceFrontendCollate(object):
    """Collate for `TextMelLoader(return_audio=True)`: same 7-tuple and ordering as TextMelCollate, but the mel
    batch is produced by ONE STFT->mel launch on the GPU from the zero-padded int16 batch (per-utterance reflect
    padding and frame counts inside the kernel; frames past an utterance are 0.0 like the reference's collate pad).
    mel / gate / output_lengths come back as device tensors — `parse_batch`'s `.cuda()` is then a no-op — so the
    DataLoader never produces a mel on the host (reference data_utils.py:42-59 + 88-137)."""

    def __init__(self, hparams, stft=None):
        if stft is None:
            import layers
            stft = layers.TacotronSTFT(hparams.filter_length, hparams.hop_length, hparams.win_length,
                                       hparams.n_mel_channels, hparams.sampling_rate, hparams.mel_fmin,
                                       hparams.mel_fmax)
        self.stft = stft
        self.hop = hparams.hop_length
        self.scale = 1.0 / hparams.max_wav_value
        self.n_frames_per_step = hparams.n_frames_per_step

    def __call__(self, batch):
        n = len(batch)
        input_lengths, order = torch.sort(torch.LongTensor([len(x[0]) for x in batch]), dim=0, descending=True)
        order = order.tolist()
        text_padded = torch.zeros(n, int(input_lengths[0]), dtype=torch.long)
        speakers = torch.zeros(n, len(batch[0][2]), dtype=torch.long)
        emotions = torch.zeros(n, len(batch[0][3]), dtype=torch.long)
        n_samples = torch.LongTensor([batch[src][1].numel() for src in order])
        wav = torch.zeros(n, int(n_samples.max()), dtype=torch.int16)
        for row, src in enumerate(order):
            text, audio, spk, emo = batch[src]
            text_padded[row, :text.size(0)] = text
            wav[row, :audio.numel()] = audio
            speakers[row] = spk
            emotions[row] = emo
        output_lengths = n_samples // self.hop + 1
        max_t = int(output_lengths.max())
        r = self.n_frames_per_step
        if max_t % r:
            max_t += r - max_t % r
        dev = torch.device('cuda')
        mel = self.stft.mel_spectrogram(wav.to(dev, non_blocking=True), lengths=n_samples, scale=self.scale)
        if mel.size(2) < max_t:
            mel = torch.nn.functional.pad(mel, (0, max_t - mel.size(2)))
        out_len_dev = output_lengths.to(dev)
        frames = torch.arange(max_t, device=dev).unsqueeze(0)
        gate = (frames >= (out_len_dev - 1).unsqueeze(1)).float()
        return text_padded, input_lengths, mel, gate, out_len_dev, speakers, emotions


class BucketBatchSampler(torch.utils.data.Sampler):
    """Length-bucketed batches for one DP rank.  Every epoch the same (seed, epoch)-keyed permutation is drawn on
    all ranks, cut into windows of `window` global batches, each window sorted by length and sliced into global
    batches; rank r takes every world_size-th item of a global batch (so all ranks get similar lengths), and the
    order of the batches is shuffled again.  Incomplete trailing batches are dropped (drop_last=True in the
    reference loader, train.py:62-65)."""

    def __init__(self, lengths, batch_size, world_size=1, rank=0, seed=1234, window=16, shuffle=True, text_lengths=None,
                 text_cap=560):
        """text_lengths / text_cap (round 4): the one-launch persistent decoder kernels took T_in <= 224 symbols then (koemo
        reaches 555); since round 6 they take 560 (t2v_decoder_train_persist_supported), which is the default cap — koemo no
        longer has a sentence above it, other corpora may.  With text lengths given, a window is sorted by (longer than the cap?, length): the few long utterances of a
        window share batches instead of pushing many batches over the cap.  persistent_hit_rate() reports the fraction of
        this rank's batches that stay inside the range."""
        self.lengths = list(lengths)
        self.text_lengths = None if text_lengths is None else list(text_lengths)
        self.text_cap = int(text_cap)
        assert self.text_lengths is None or len(self.text_lengths) == len(self.lengths)
        self.batch_size, self.world_size, self.rank = int(batch_size), int(world_size), int(rank)
        self.seed, self.window, self.shuffle = int(seed), int(window), bool(shuffle)
        self.epoch = 0
        self.global_batch = self.batch_size * self.world_size
        self.n_batches = len(self.lengths) // self.global_batch

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def __len__(self):
        return self.n_batches

    def _key(self, i):
        if self.text_lengths is None:
            return (0, self.lengths[i])
        return (1 if self.text_lengths[i] > self.text_cap else 0, self.lengths[i])

    def persistent_hit_rate(self):
        """fraction of this rank's batches (current epoch) whose longest text fits the persistent decoder kernels"""
        if self.text_lengths is None:
            return None
        tot = hit = 0
        for b in self:
            tot += 1
            hit += max(self.text_lengths[i] for i in b) <= self.text_cap
        return hit / max(1, tot)

    def __iter__(self):
        g = torch.Generator()
        g.manual_seed(self.seed + self.epoch)
        n = len(self.lengths)
        perm = torch.randperm(n, generator=g).tolist() if self.shuffle else list(range(n))
        span = self.global_batch * self.window
        batches = []
        for lo in range(0, n, span):
            chunk = sorted(perm[lo:lo + span], key=self._key)
            for b in range(0, len(chunk) - self.global_batch + 1, self.global_batch):
                batches.append(chunk[b:b + self.global_batch])
        batches = batches[:self.n_batches]
        if self.shuffle:
            order = torch.randperm(len(batches), generator=g).tolist()
            batches = [batches[i] for i in order]
        for gb in batches:
            yield gb[self.rank::self.world_size]
