"""Golden LOSS TRAJECTORY of the reference's training loop: the small model in train mode, the reference's own
MultiResolutionSTFTLoss (main_config.yaml factors) and torch.optim.Adam exactly as train.py:83 builds it (lr 3e-4, betas (0.9, 0.999)),
30 steps of solver.py:296-305 + 602-605 on ONE fixed batch, all in fp32 on the CPU.  tests/test_gpu_train.py holds the HIP training
loop (fp16 activation storage, fused Adam) to this trajectory.  Runs only in the build container (imports /root/reference):
    python -B oracle/make_golden_train.py
Nothing of the reference is copied: the fixture is 30 x {sc, mag} loss values and the seeds that reproduce the inputs."""
import json
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, '..', 'tests', 'golden')
sys.path.insert(0, '/root/reference')
SMALL_CFG = dict(channels=16, nfft=256, hop_length=32, lr_sr=4000, hr_sr=16000, enc_freq_attn=0)
STEPS, LR = 30, 3e-4


def seeded(shape, seed):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def main():
    torch.set_num_threads(8)
    from src.models.aero import Aero
    import src.models.stft_loss as ref_loss
    real_stft = torch.stft

    def stft_compat(x, n_fft, hop_length=None, win_length=None, window=None, **kw):       # stft_loss.py:22 predates return_complex
        return torch.view_as_real(real_stft(x, n_fft, hop_length, win_length, window, return_complex=True, **kw))
    torch.manual_seed(21)
    m = Aero(**SMALL_CFG).train()
    x, hr = seeded((2, 1, 2003), 1), 0.1 * seeded((2, 1, 8012), 2)
    crit = ref_loss.MultiResolutionSTFTLoss(factor_sc=0.5, factor_mag=0.5)
    opt = torch.optim.Adam(m.parameters(), lr=LR, betas=(0.9, 0.999))
    traj = []
    for i in range(STEPS):
        pr = m(x)
        torch.stft = stft_compat                                   # (only around the loss: the model's own STFT passes return_complex itself)
        try:
            sc, mag = crit(pr.squeeze(1), hr.squeeze(1))
        finally:
            torch.stft = real_stft
        loss = sc + mag
        opt.zero_grad()
        loss.backward()
        opt.step()
        traj.append((float(sc.detach()), float(mag.detach())))
        print(i, traj[-1], flush=True)
    np.savez_compressed(os.path.join(OUT, 'train_small_trajectory.npz'), loss=np.array(traj, dtype=np.float64))
    mp = os.path.join(OUT, 'meta.json')
    meta = json.load(open(mp))
    meta['train_small_trajectory'] = {'model_seed': 21, 'x_seed': 1, 'hr_seed': 2, 'hr_scale': 0.1, 'L': 2003, 'steps': STEPS, 'lr': LR,
                                      'betas': [0.9, 0.999], 'factor_sc': 0.5, 'factor_mag': 0.5}
    json.dump(meta, open(mp, 'w'), indent=1)


if __name__ == '__main__':
    main()
