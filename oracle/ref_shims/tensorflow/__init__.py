"""Shim for `tf.contrib.training.HParams` + `tf.logging` (hparams.py:1,6,120-124)."""
import types


class HParams(object):
    def __init__(self, **kw):
        self.__dict__['_v'] = dict(kw)

    def __getattr__(self, k):
        try:
            return self.__dict__['_v'][k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self._v[k] = v

    def values(self):
        return dict(self._v)

    def parse(self, s):
        for item in s.split(','):
            if not item.strip():
                continue
            k, v = item.split('=', 1)
            k = k.strip()
            cur = self._v[k]
            if isinstance(cur, bool):
                v = v.strip().lower() in ('true', '1')
            elif isinstance(cur, int):
                v = int(v)
            elif isinstance(cur, float):
                v = float(v)
            elif isinstance(cur, (list, tuple)):
                raise ValueError('list-valued hparam override unsupported: ' + k)
            self._v[k] = v
        return self


contrib = types.SimpleNamespace(training=types.SimpleNamespace(HParams=HParams))
logging = types.SimpleNamespace(info=lambda *a, **k: None)
