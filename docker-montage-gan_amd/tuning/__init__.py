"""Tuned solver selection for the convolution library (MIOpen) on MI355X.

`miopen/` holds MIOpen's *user find-db* and *user perf-db* text files written by one exhaustive search
(`python bench.py --miopen-find --miopen-db-out DIR`, ~14 minutes on a fresh MI355X) over the convolution
problems of the benchmark workload (256x256 RGBA, bf16 NHWC blocks + fp32 NCHW blocks) at per-GPU batch 32 (BASELINE configs 2-4;
140 problems, round 1) and, appended by a second search seeded with those files, at per-GPU batch 4 (BASELINE config 5: global batch
32 over 8 GPUs; +138 problems, round 2: 470 -> 419 ms per eager step at batch 4) and 16 and 8 (BASELINE config 3 and the sizes
between; +123 problems: 467 -> 414 ms and 374 -> 309 ms per recorded step).  Other batches miss the db and get MIOpen's heuristics.  They are
measurements of MIOpen's own solvers on gfx950 -- which solver is fastest for which problem -- not code.

`use_tuned_miopen_db()` points MIOpen at a private copy of those files (MIOPEN_USER_DB_PATH).  MIOpen's
immediate mode (``torch.backends.cudnn.benchmark = False``) consults the find-db first: a problem found there gets
its measured-fastest solver at once, a problem that is not falls back to the built-in heuristics -- no search
either way, so the first step costs kernel compilation only.  With ``cudnn.benchmark = True`` (the reference's
setting, training_loop.py:124) pass ``find_mode='FAST'`` for the same behaviour through MIOpen's find API.
It must run before the first convolution of the process.
"""
import atexit
import os
import shutil
import tempfile

DB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'miopen')


def tuned_db_files():
    return sorted(f for f in os.listdir(DB_DIR) if f.endswith('.txt')) if os.path.isdir(DB_DIR) else []


def use_tuned_miopen_db(find_mode=None):
    """Returns the directory MIOpen was pointed at, or None when no tuned files ship with this tree."""
    files = tuned_db_files()
    if not files:
        return None
    # a private, writable copy per process: MIOpen locks and may append to its user databases
    dst = tempfile.mkdtemp(prefix='montage_miopen_db_')
    atexit.register(shutil.rmtree, dst, ignore_errors=True)
    for f in files:
        shutil.copy(os.path.join(DB_DIR, f), os.path.join(dst, f))
    os.environ['MIOPEN_USER_DB_PATH'] = dst
    if find_mode:
        os.environ['MIOPEN_FIND_MODE'] = find_mode
    return dst

