"""Container-only loader for the read-only reference (TEST INFRASTRUCTURE).

Imports `/root/reference` on CPU through `oracle/ref_shims` so that
`oracle/gen_golden.py` can run the reference's own code and write fixtures to
`tests/golden/`.  Never imported by the product, by `-m gpu` tests, by
`bench.py` or by `smoke()`; `/root/reference` does not exist on the GPU box.
Patch list follows SURVEY.md Appendix A.
"""
import os
import sys

REF = os.environ.get('T2V_REFERENCE', '/root/reference')


def available():
    return os.path.isdir(REF) and os.path.isfile(os.path.join(REF, 'model.py'))


def load():
    """Returns dict of reference modules; idempotent."""
    import torch
    sys.dont_write_bytecode = True
    here = os.path.dirname(os.path.abspath(__file__))
    shims = os.path.join(here, 'ref_shims')
    for p in (REF, shims):
        if p in sys.path:
            sys.path.remove(p)
    sys.path.insert(0, REF)
    sys.path.insert(0, shims)
    # CUDA-isms -> identity on CPU (train.py:81, CoordConv.py:62-65, utils.py:27-32)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    import utils as r_utils
    import model as r_model
    import modules as r_modules
    import layers as r_layers
    import loss_function as r_loss
    import data_utils as r_data
    import hparams as r_hparams
    import text as r_text
    import stft as r_stft

    def _mask(lengths):  # utils.py:9-13 with a bool mask (modern torch)
        max_len = int(torch.max(lengths).item())
        ids = torch.arange(0, max_len, dtype=torch.long)
        return ids < lengths.unsqueeze(1)

    r_utils.get_mask_from_lengths = _mask
    r_model.get_mask_from_lengths = _mask
    return dict(utils=r_utils, model=r_model, modules=r_modules, layers=r_layers,
                loss=r_loss, data=r_data, hparams=r_hparams, text=r_text, stft=r_stft)
