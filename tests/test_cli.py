"""train.py keeps the reference's command line (reference train.py:252-285): same flags, same defaults."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_train_cli_flags():
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tacotron2-vae_amd', 'train.py'), '--help'],
                         capture_output=True, text=True, timeout=300, cwd=os.path.join(ROOT, 'tacotron2-vae_amd'))
    assert out.returncode == 0, out.stderr
    for flag in ('-o', '--output_directory', '-l', '--log_directory', '-c', '--checkpoint_path', '--warm_start',
                 '--n_gpus', '--rank', '--group_name', '--hparams'):
        assert flag in out.stdout, flag


def test_bench_cli_flags():
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--help'], capture_output=True, text=True,
                         timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr
    for flag in ('--gpus', '--steps', '--warmup', '--bf16', '--koemo'):
        assert flag in out.stdout, flag
