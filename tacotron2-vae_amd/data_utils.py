"""Dataset + collate with the reference's batch layout (reference data_utils.py:11-137)."""
import random

import numpy as np
import torch
import torch.utils.data

from utils import load_filepaths_and_text, load_wav_to_torch


class TextMelLoader(torch.utils.data.Dataset):
    """filelist line `path|text|speaker|emotion` → (ids IntTensor, mel (80,T), one-hot speaker,
    one-hot emotion).  The list is shuffled once with random.seed(1234) like data_utils.py:29-30."""

    def __init__(self, audiopaths_and_text, hparams, stft=None):
        self.audiopaths_and_text = load_filepaths_and_text(audiopaths_and_text)
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
        return (self.get_text(text), self.get_mel(path), self.get_speaker(speaker), self.get_emotion(emotion))

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
