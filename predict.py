"""predict.py -- enhance one low-resolution wav with the MI355X-native AERO generator.

    python predict.py dset=<dset> experiment=<experiment> +filename=<in.wav> +output=<out dir> [checkpoint_file=...]

Same arguments and behaviour as the reference's predict.py:41-99 (10-second independent chunks, concatenation,
clip-safe normalisation on write); configuration is read from conf/ by aero_amd.config (hydra is optional).
"""
import logging
import os
import sys
import time
from pathlib import Path

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from aero_amd import audio_io, enhance  # noqa: E402
from aero_amd.config import load_config  # noqa: E402

logger = logging.getLogger('predict')


def main(argv=None):
    logging.basicConfig(level=logging.INFO)
    args = load_config(os.path.join(ROOT, 'conf'), argv if argv is not None else sys.argv[1:])
    model = enhance.load_generator(args, device='cuda')
    lr_sig, sr = audio_io.load(args.filename)
    if args.experiment.upsample:                                   # predict.py:55-57 (no aero config sets it; host-side sinc resampling)
        lr_sig = audio_io.resample(lr_sig, sr, args.experiment.hr_sr)
        sr = args.experiment.hr_sr
    logger.info(f'lr wav shape: {tuple(lr_sig.shape)}')
    t0 = time.time()
    pr = enhance.predict_signal(model, lr_sig, sr)
    logger.info(f'prediction duration: {time.time() - t0}')
    logger.info(f'pr wav shape: {tuple(pr.shape)}')
    os.makedirs(args.output, exist_ok=True)
    out = os.path.join(args.output, Path(args.filename).stem + '_pr.wav')
    logger.info(f'saving to: {out}, with sample_rate: {args.experiment.hr_sr}')
    enhance.write(pr, out, args.experiment.hr_sr)
    return out


if __name__ == '__main__':
    main()
