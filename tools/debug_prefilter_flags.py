"""Diagnostic: which users of a prefiltered call take the exact fallback, and why (log counts, exact candidates above tau).
Reads the call's workspace with the layout of make_plan (tgcn_score_fused.hip)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textgcn_amd import scoring  # noqa: E402


def plan(B, I):
    al = lambda x: (x + 255) & ~255
    S = min(32, max(1, I // 512))
    ips = (((I + S - 1) // S + 63) // 64) * 64
    S = (I + ips - 1) // ips
    cap2 = max(32, 1024 // (2 * S))
    m = (I + 31) // 32
    m_ld = (m + 3) & ~3
    o, off = 0, {}
    for name, size in (('sample', B * m_ld * 4), ('tauv', B * 10 * 4), ('taui', B * 10 * 8), ('tau', B * 4), ('taulo', B * 4),
                       ('npart', 1024 * 4), ('logs', B * S * 2 * cap2 * 8), ('counts', B * S * 2 * 4), ('parts', B * 32 * 64 * 8),
                       ('flags', (B + 1) * 4), ('done', B * 4)):
        off[name] = o
        o += al(size)
    return S, cap2, off, o


def main():
    dev = torch.device('cuda:0')
    B, I, d, k = 2048, 60000, int(sys.argv[1]) if len(sys.argv) > 1 else 128, 40
    g = torch.Generator().manual_seed(0)
    for trial in range(6):
        ue = (torch.randn(B, d, generator=g) * 0.1).to(dev)
        ie = (torch.randn(I, d, generator=g) * 0.1).to(dev)
        for mode in (False, True):
            scoring.score_topk(ue, ie, k, prefilter=mode)
            torch.cuda.synchronize()
            ws = scoring._WORKSPACE[(dev, 0)]
            S, cap2, off, total = plan(B, I)
            raw = ws[:total].cpu().numpy()
            flags = raw[off['flags']:off['flags'] + (B + 1) * 4].view(np.int32)
            n_flag = int(flags[0])
            tau = raw[off['tau']:off['tau'] + B * 4].view(np.float32)
            taulo = raw[off['taulo']:off['taulo'] + B * 4].view(np.float32)
            counts = raw[off['counts']:off['counts'] + B * S * 2 * 4].view(np.int32).reshape(B, S * 2)
            print(f'trial {trial} prefilter={mode}: flagged {n_flag}; logged per user mean {counts.sum(1).mean():.1f} max {counts.sum(1).max()} '
                  f'max lane log {counts.max()} (cap {cap2})')
            if n_flag:
                s = scoring.score_dense(ue, ie)
                for u in flags[1:1 + min(n_flag, 4)]:
                    above = int((s[u] > float(tau[u])).sum())
                    print(f'   user {u}: tau {tau[u]:.5f} tau_lo {taulo[u]:.5f} exact>tau {above} logged {counts[u].sum()} max lane log {counts[u].max()}')
                    if mode:
                        logs = raw[off['logs']:off['logs'] + B * S * 2 * cap2 * 8].view(np.float32).reshape(B, S * 2, cap2, 2)
                        kept = 0
                        for sg in range(S * 2):
                            c = min(counts[u, sg], cap2)
                            items = logs[u, sg, :c, 1].view(np.int32)
                            kept += int((items != 2147483647).sum())
                        print(f'      entries kept after rescoring: {kept}')


if __name__ == '__main__':
    main()
