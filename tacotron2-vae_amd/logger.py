"""TensorBoard logging for the training loop: the tags and call signatures of the reference's
`Tacotron2Logger` (reference logger.py:9-56) on a self-contained event-file writer.

The reference derives from tensorboardX.SummaryWriter; neither tensorboardX nor tensorboard is a
dependency here, so this module writes the TFRecord/`Event` wire format itself (scalars, histograms,
PNG images) — `tensorboard --logdir` reads the result.  Images need matplotlib (Agg); without it the
scalar and histogram streams are still written and the image calls are skipped.
"""
import os
import random
import socket
import struct
import time
import zlib

import numpy as np

# --------------------------------------------------------------------------- TFRecord framing
_CRC_TABLE = None


def _crc32c(data):
    """CRC-32C (Castagnoli), table driven."""
    global _CRC_TABLE
    if _CRC_TABLE is None:
        tab = []
        for n in range(256):
            c = n
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            tab.append(c)
        _CRC_TABLE = tab
    crc = 0xFFFFFFFF
    for b in data:
        crc = _CRC_TABLE[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def _masked_crc(data):
    c = _crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _record(payload):
    head = struct.pack('<Q', len(payload))
    return head + struct.pack('<I', _masked_crc(head)) + payload + struct.pack('<I', _masked_crc(payload))


# --------------------------------------------------------------------------- protobuf wire format
def _varint(n):
    out = bytearray()
    n &= (1 << 64) - 1
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _key(field, wire):
    return _varint((field << 3) | wire)


def _f_double(field, v):
    return _key(field, 1) + struct.pack('<d', float(v))


def _f_float(field, v):
    return _key(field, 5) + struct.pack('<f', float(v))


def _f_int(field, v):
    return _key(field, 0) + _varint(int(v))


def _f_bytes(field, b):
    if isinstance(b, str):
        b = b.encode('utf-8')
    return _key(field, 2) + _varint(len(b)) + bytes(b)


def _f_packed_doubles(field, values):
    body = struct.pack('<%dd' % len(values), *[float(v) for v in values])
    return _key(field, 2) + _varint(len(body)) + body


def _event(step, summary_value, wall_time=None):
    """Event{wall_time=1, step=2, summary=5{value=1{...}}}"""
    summary = _f_bytes(1, summary_value)
    return (_f_double(1, time.time() if wall_time is None else wall_time) + _f_int(2, step) + _f_bytes(5, summary))


def _png(rgb):
    """Minimal PNG encoder for an (H, W, 3) uint8 array."""
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w, _ = rgb.shape
    raw = b''.join(b'\x00' + rgb[y].tobytes() for y in range(h))

    def chunk(tag, data):
        return struct.pack('>I', len(data)) + tag + data + struct.pack('>I', zlib.crc32(tag + data) & 0xFFFFFFFF)
    return (b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, 8, 2, 0, 0, 0)) +
            chunk(b'IDAT', zlib.compress(raw, 6)) + chunk(b'IEND', b''))


class EventFileWriter(object):
    """`events.out.tfevents.<time>.<host>` in `logdir`, one record per add_* call."""

    def __init__(self, logdir):
        os.makedirs(logdir, exist_ok=True)
        self.logdir = logdir
        self.path = os.path.join(logdir, 'events.out.tfevents.%010d.%s' % (int(time.time()), socket.gethostname()))
        self._f = open(self.path, 'wb')
        # first record: file version
        self._f.write(_record(_f_double(1, time.time()) + _f_bytes(3, 'brain.Event:2')))
        self._f.flush()

    def _write(self, payload):
        self._f.write(_record(payload))

    def add_scalar(self, tag, value, step):
        """Summary.Value{tag=1, simple_value=2}"""
        if hasattr(value, 'item'):
            value = value.item()
        self._write(_event(step, _f_bytes(1, tag) + _f_float(2, value)))

    def add_histogram(self, tag, values, step, bins=30):
        """Summary.Value{tag=1, histo=5{min=1,max=2,num=3,sum=4,sum_squares=5,bucket_limit=6,bucket=7}}"""
        v = np.asarray(values, dtype=np.float64).reshape(-1)
        if v.size == 0:
            return
        lo, hi = float(v.min()), float(v.max())
        if hi <= lo:
            hi = lo + 1e-12
        counts, edges = np.histogram(v, bins=bins, range=(lo, hi))
        histo = (_f_double(1, lo) + _f_double(2, hi) + _f_double(3, v.size) + _f_double(4, v.sum()) +
                 _f_double(5, (v * v).sum()) + _f_packed_doubles(6, edges[1:]) + _f_packed_doubles(7, counts))
        self._write(_event(step, _f_bytes(1, tag) + _f_bytes(5, histo)))

    def add_image(self, tag, img_hwc, step):
        """Summary.Value{tag=1, image=4{height=1,width=2,colorspace=3,encoded_image_string=4}}; (H,W,3) uint8."""
        img = np.asarray(img_hwc)
        if img.ndim != 3 or img.shape[2] != 3:
            raise ValueError("add_image expects an (H, W, 3) uint8 array")
        image = _f_int(1, img.shape[0]) + _f_int(2, img.shape[1]) + _f_int(3, 3) + _f_bytes(4, _png(img))
        self._write(_event(step, _f_bytes(1, tag) + _f_bytes(4, image)))

    def flush(self):
        self._f.flush()

    def close(self):
        if not self._f.closed:
            self._f.flush()
            self._f.close()


def read_events(path):
    """Decode an event file written above into [(step, tag, kind, value)] (kind: 'scalar'|'histo'|'image');
    verifies both CRCs of every record.  Used by the tests and handy for loss-curve comparisons."""
    def parse(buf):
        i, out = 0, []
        while i < len(buf):
            key, shift = 0, 0
            while True:
                b = buf[i]; i += 1
                key |= (b & 0x7F) << shift
                shift += 7
                if not b & 0x80:
                    break
            field, wire = key >> 3, key & 7
            if wire == 0:
                val, shift = 0, 0
                while True:
                    b = buf[i]; i += 1
                    val |= (b & 0x7F) << shift
                    shift += 7
                    if not b & 0x80:
                        break
            elif wire == 1:
                val = struct.unpack('<d', buf[i:i + 8])[0]; i += 8
            elif wire == 5:
                val = struct.unpack('<f', buf[i:i + 4])[0]; i += 4
            elif wire == 2:
                n, shift = 0, 0
                while True:
                    b = buf[i]; i += 1
                    n |= (b & 0x7F) << shift
                    shift += 7
                    if not b & 0x80:
                        break
                val = bytes(buf[i:i + n]); i += n
            else:
                raise ValueError("unsupported wire type %d" % wire)
            out.append((field, wire, val))
        return out

    events = []
    with open(path, 'rb') as f:
        data = f.read()
    pos = 0
    while pos < len(data):
        head = data[pos:pos + 8]
        n = struct.unpack('<Q', head)[0]
        if struct.unpack('<I', data[pos + 8:pos + 12])[0] != _masked_crc(head):
            raise ValueError("bad length CRC at %d" % pos)
        payload = data[pos + 12:pos + 12 + n]
        if struct.unpack('<I', data[pos + 12 + n:pos + 16 + n])[0] != _masked_crc(payload):
            raise ValueError("bad payload CRC at %d" % pos)
        pos += 16 + n
        step, summary = 0, None
        for field, _, val in parse(payload):
            if field == 2:
                step = val
            elif field == 5:
                summary = val
        if summary is None:
            continue
        for field, _, value_msg in parse(summary):
            if field != 1:
                continue
            tag, kind, value = None, None, None
            for f2, _, v2 in parse(value_msg):
                if f2 == 1:
                    tag = v2.decode('utf-8')
                elif f2 == 2:
                    kind, value = 'scalar', v2
                elif f2 == 5:
                    kind, value = 'histo', {k: v for k, _, v in parse(v2) if k <= 5}
                elif f2 == 4:
                    d = {k: v for k, _, v in parse(v2)}
                    kind, value = 'image', (d.get(1), d.get(2), d.get(4, b'')[:8])
            events.append((step, tag, kind, value))
    return events


# --------------------------------------------------------------------------- plots (reference plotting_utils.py)
def _figure_to_numpy(fig):
    fig.canvas.draw()
    buf = np.asarray(fig.canvas.buffer_rgba())
    return np.ascontiguousarray(buf[:, :, :3])


def _plots():
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        return plt
    except Exception:       # no matplotlib: scalars/histograms only
        return None


def plot_alignment_to_numpy(alignment, info=None):
    plt = _plots()
    fig, ax = plt.subplots(figsize=(6, 4))
    im = ax.imshow(alignment, aspect='auto', origin='lower', interpolation='none')
    fig.colorbar(im, ax=ax)
    ax.set_xlabel('Decoder timestep' + ('\n\n' + info if info else ''))
    ax.set_ylabel('Encoder timestep')
    fig.tight_layout()
    data = _figure_to_numpy(fig)
    plt.close(fig)
    return data


def plot_spectrogram_to_numpy(spectrogram):
    plt = _plots()
    fig, ax = plt.subplots(figsize=(12, 3))
    im = ax.imshow(spectrogram, aspect='auto', origin='lower', interpolation='none')
    fig.colorbar(im, ax=ax)
    ax.set_xlabel('Frames')
    ax.set_ylabel('Channels')
    fig.tight_layout()
    data = _figure_to_numpy(fig)
    plt.close(fig)
    return data


def plot_gate_outputs_to_numpy(gate_targets, gate_outputs):
    plt = _plots()
    fig, ax = plt.subplots(figsize=(12, 3))
    ax.scatter(range(len(gate_targets)), gate_targets, alpha=0.5, color='green', marker='+', s=1, label='target')
    ax.scatter(range(len(gate_outputs)), gate_outputs, alpha=0.5, color='red', marker='.', s=1, label='predicted')
    ax.set_xlabel('Frames (Green target, Red predicted)')
    ax.set_ylabel('Gate State')
    fig.tight_layout()
    data = _figure_to_numpy(fig)
    plt.close(fig)
    return data


def plot_scatter(mus, emotions):
    """first two latent dimensions coloured by emotion label (neu, sad, ang, hap)"""
    plt = _plots()
    mus = np.asarray(mus)
    y = np.argmax(np.asarray(emotions), 1)
    fig, ax = plt.subplots(figsize=(12, 12))
    for i, (c, label) in enumerate(zip(('r', 'b', 'g', 'y'), ('neu', 'sad', 'ang', 'hap'))):
        ax.scatter(mus[y == i, 0], mus[y == i, 1], c=c, label=label, alpha=0.5)
    ax.legend(loc='upper left')
    data = _figure_to_numpy(fig)
    plt.close(fig)
    return data


class Tacotron2Logger(EventFileWriter):
    """Same method names, argument order and tags as reference logger.py:9-56."""

    def __init__(self, logdir):
        super(Tacotron2Logger, self).__init__(logdir)

    def log_training(self, reduced_loss, grad_norm, learning_rate, duration, recon_loss, kl_div, kl_weight,
                     iteration):
        self.add_scalar("training.loss", reduced_loss, iteration)
        self.add_scalar("grad.norm", grad_norm, iteration)
        self.add_scalar("learning.rate", learning_rate, iteration)
        self.add_scalar("duration", duration, iteration)
        self.add_scalar("kl_div", kl_div, iteration)
        self.add_scalar("kl_weight", kl_weight, iteration)
        self.add_scalar("recon_loss", recon_loss, iteration)
        self.flush()

    def log_validation(self, reduced_loss, model, y, y_pred, iteration):
        self.add_scalar("validation.loss", reduced_loss, iteration)
        _, mel_outputs, gate_outputs, alignments, mus, _, _, emotions = y_pred
        mel_targets, gate_targets = y
        for tag, value in model.named_parameters():
            self.add_histogram(tag.replace('.', '/'), value.detach().float().cpu().numpy(), iteration)
        if _plots() is not None:
            import torch
            idx = random.randint(0, alignments.size(0) - 1)
            self.add_image("alignment", plot_alignment_to_numpy(alignments[idx].detach().cpu().numpy().T), iteration)
            self.add_image("mel_target", plot_spectrogram_to_numpy(mel_targets[idx].detach().cpu().numpy()), iteration)
            self.add_image("mel_predicted", plot_spectrogram_to_numpy(mel_outputs[idx].detach().cpu().numpy()),
                           iteration)
            self.add_image("gate", plot_gate_outputs_to_numpy(
                gate_targets[idx].detach().cpu().numpy(),
                torch.sigmoid(gate_outputs[idx]).detach().cpu().numpy()), iteration)
            self.add_image("latent_dim", plot_scatter(mus.detach().cpu().numpy(), emotions.detach().cpu().numpy()),
                           iteration)
        self.flush()
