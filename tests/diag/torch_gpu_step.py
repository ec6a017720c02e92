"""Diagnostic (GPU box): the SPEC's own train step -- oracle/colvo_spec.py, stock torch ops, MIOpen convolutions -- on the GPU, timed
beside the hand-written path at the same shape.  BASELINE.json has no published number and /root/reference no code, so this is the
only "what would the reference's PyTorch code do on this card" figure there is: the oracle modules moved to cuda:0, channels_last,
fp32 and bf16 autocast, torch.optim.Adam.  Lives under tests/ because it imports the oracle (test infrastructure); nothing in the
product or in bench.py's timed region uses it.

   python tests/diag/torch_gpu_step.py [pairs=8] [steps=30]        -> one JSON line"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from coivo_amd import synth  # noqa: E402
from oracle import colvo_spec as S  # noqa: E402


def run(pairs, steps, amp, benchmark):
    torch.backends.cudnn.benchmark = benchmark
    dev = torch.device("cuda:0")
    dn, pn = S.make_models(0)
    dn, pn = dn.to(dev).to(memory_format=torch.channels_last), pn.to(dev).to(memory_format=torch.channels_last)
    opt = torch.optim.Adam(list(dn.parameters()) + list(pn.parameters()), **S.ADAM_KW)
    b = synth.make_batch(pairs, 256, 320, seed=1234, device=dev)
    tgt, ref, K = b["tgt"], b["ref"], b["K"]

    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            d = dn(torch.cat([tgt, ref]).contiguous(memory_format=torch.channels_last))
            d_t, d_r = d[:pairs].float(), d[pairs:].float()
            pose, a, bb = pn(tgt, ref, d_t, d_r)
        loss = S.photometric_loss(tgt, ref, d_t, pose.float(), K, a.float(), bb.float())
        loss.backward()
        opt.step()
        return loss

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(steps):
        step()
        ev[i + 1].record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps * 1e3
    per = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(steps))
    return {"ms_per_step_wall": round(wall, 3), "ms_per_step_hipevent_median": round(per[len(per) // 2], 3),
            "pairs_per_s": round(pairs / (wall * 1e-3), 1)}


def main():
    pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    out = {"what": "oracle/colvo_spec.py train step on cuda:0 (stock torch ops, MIOpen convolutions, torch.optim.Adam), channels_last",
           "pairs": pairs, "shape": "320x256", "torch": torch.__version__}
    for tag, amp, bm in (("fp32", False, False), ("bf16_autocast", True, False), ("bf16_autocast_miopen_find", True, True)):
        try:
            out[tag] = run(pairs, steps, amp, bm)
        except Exception as e:          # noqa: BLE001
            out[tag] = {"error": f"{type(e).__name__}: {e}"[:200]}
        print(tag, out[tag], file=sys.stderr, flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
