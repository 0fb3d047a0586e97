"""test.py -- evaluate the MI355X-native AERO generator on (lr, hr) wav pairs: mean log-spectral distance.

    python test.py dset=<dset> experiment=<experiment> checkpoint_file=<ckpt> [+lr_dir=<dir> +hr_dir=<dir>]

Counterpart of the reference's test.py:41-52 + src/evaluate.py: pairs are matched by file stem, every pair goes through
Aero.forward once (pr, pr_spec, lr_spec) and model._spec(hr, scale=True); files are sharded over ranks (clip i -> rank
i mod W) and the LSD sums are all-reduced.  `dset.test` may point to a directory holding lr.json / hr.json
([[path, n_samples], ...], as written by the reference's data_prep scripts) or `+lr_dir/+hr_dir` may name two wav folders.
"""
import json
import logging
import os
import sys
from pathlib import Path

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from aero_amd import audio_io, distrib, enhance, evaluate  # noqa: E402
from aero_amd.config import load_config  # noqa: E402

logger = logging.getLogger('test')


def _listing(args):
    """sorted (lr_path, hr_path) pairs matched by stem (datasets.py:24-37)."""
    if args.get('lr_dir') and args.get('hr_dir'):
        lr = sorted(str(p) for p in Path(args.lr_dir).glob('*.wav'))
        hr = sorted(str(p) for p in Path(args.hr_dir).glob('*.wav'))
    else:
        base = Path(str(args.dset.test))
        lr = sorted(p for p, _ in json.load(open(base / 'lr.json')))
        hr = sorted(p for p, _ in json.load(open(base / 'hr.json')))
    lr_by_stem = {Path(p).stem: p for p in lr}
    pairs = [(lr_by_stem[Path(p).stem], p) for p in hr if Path(p).stem in lr_by_stem]
    if not pairs:
        raise RuntimeError('no (lr, hr) pairs with matching file names')
    return pairs


def main(argv=None):
    logging.basicConfig(level=logging.INFO)
    args = load_config(os.path.join(ROOT, 'conf'), argv if argv is not None else sys.argv[1:])
    distrib.init_from_env()
    model = enhance.load_generator(args, device='cuda')
    files = _listing(args)

    mine = [files[i] for i in distrib.shard_indices(len(files))]     # file i -> rank i mod W, sharded BEFORE any decoding

    def pairs():
        for lr_path, hr_path in mine:
            lr, _ = audio_io.load(lr_path)
            hr, _ = audio_io.load(hr_path)
            yield lr[:1].unsqueeze(0), hr[:1].unsqueeze(0)
    total, count, _ = evaluate.evaluate(model, pairs(), device='cuda')
    lsd = distrib.average([total / max(count, 1)], count)[0]          # file-weighted mean over ranks (distrib.py:43-55)
    logger.info(f'Done evaluation.  LSD={lsd} , VISQOL=0 (external binary not configured), files={len(files)}')
    return lsd


if __name__ == '__main__':
    main()
