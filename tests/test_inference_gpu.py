"""-m gpu parity of the inference side (trajectory -> back-projection -> stitched cloud) with the oracle."""
import pytest
import torch

from coivo_amd import synth
from tests.gpu_util import dev, to_dev

pytestmark = pytest.mark.gpu

# fp32 point coordinates: the oracle's bmm and the kernel's fused multiply-adds round differently; 1e-5 of the cloud's extent
POINT_TOL = 1e-5


def _scene(B, H, W, seed):
    from oracle import colvo_spec as S
    g = torch.Generator().manual_seed(seed)
    depth = 0.3 + 4.0 * torch.rand(B, 1, H, W, generator=g)
    K = synth.intrinsics(B, H, W)
    rel = torch.cat([0.05 * torch.randn(B, 3, generator=g), 0.1 * torch.randn(B, 3, generator=g)], dim=1)
    M = S.integrate_trajectory(rel.double())[1:].float()
    return depth, K, M, rel


@pytest.mark.parametrize("B,H,W", [(2, 32, 64), (3, 17, 23), (1, 256, 320), (8, 256, 320)])
def test_backproject_matches_oracle(B, H, W):
    from coivo_amd import inference as I
    from oracle import colvo_spec as S
    depth, K, M, _ = _scene(B, H, W, 50 + B)
    want = S.backproject(depth.double(), K.double(), M.double())
    got = I.backproject(depth.to(dev()), K.to(dev()), M.to(dev()))
    assert got.shape == (B, H * W, 3)
    assert (got.cpu().double() - want).abs().max().item() < POINT_TOL * want.abs().max().item()


@pytest.mark.parametrize("N,H,W,stride", [(2, 32, 64, 1), (3, 17, 23, 2), (4, 64, 96, 3), (8, 256, 320, 4), (2, 5, 7, 9)])
def test_stitched_cloud_matches_oracle(N, H, W, stride):
    """Same points in the same order; a tenth of the pixels sit at or beyond max_depth and must be dropped."""
    from coivo_amd import inference as I
    from oracle import colvo_spec as S
    depth, K, M, _ = _scene(N, H, W, 60 + N)
    g = torch.Generator().manual_seed(7)
    far = torch.rand(depth.shape, generator=g) < 0.1
    depth[far] = S.MAX_DEPTH + torch.rand(int(far.sum()), generator=g)
    depth.view(-1)[0] = S.MAX_DEPTH                      # exactly at the limit: dropped
    want = S.stitch_point_cloud(depth.double(), K.double(), M.double(), stride=stride)
    got = I.stitch_point_cloud(depth.to(dev()), K.to(dev()), M.to(dev()), stride=stride)
    assert got.shape == want.shape
    assert (got.cpu().double() - want).abs().max().item() < POINT_TOL * want.abs().max().item()
    none = I.stitch_point_cloud(torch.full_like(depth, 20.0).to(dev()), K.to(dev()), M.to(dev()), stride=stride)
    assert none.shape == (0, 3)


def test_trajectory_integration_matches_oracle():
    from coivo_amd import inference as I
    from oracle import colvo_spec as S
    _, _, _, rel = _scene(12, 8, 8, 70)
    want = S.integrate_trajectory(rel.double())
    got = I.integrate_trajectory(rel.to(dev()))
    assert got.dtype == torch.float64 and got.shape == (13, 4, 4)
    assert (got - want).abs().max().item() < 1e-12


def test_reconstruct_sequence_end_to_end():
    """Networks + trajectory + stitching on a 5-frame sequence against the same pipeline on the oracle's networks."""
    from coivo_amd import inference as I, nn as hnn
    from oracle import colvo_spec as S
    dn_o, pn_o = S.make_models(31)
    dn, pn = hnn.DepthNet(), hnn.PoseNet()
    dn.load_state_dict(dn_o.state_dict())
    pn.load_state_dict(pn_o.state_dict())
    n, H, W = 5, 64, 96
    b = synth.make_batch(n, H, W, seed=31)
    frames, K = b["tgt"], b["K"]
    with torch.no_grad():
        d_o = dn_o(frames)
        p_o, _, _ = pn_o(frames[:-1], frames[1:], d_o[:-1], d_o[1:])
        traj_o = S.integrate_trajectory(p_o.double())
        cloud_o = S.stitch_point_cloud(d_o, K, traj_o.float(), stride=4)
    rec = I.reconstruct_sequence(dn, pn, frames.to(dev()), K.to(dev()), stride=4, chunk=2)
    assert (rec.depths.cpu() - d_o).abs().max().item() < 1e-4
    assert (rec.rel_poses.cpu() - p_o).abs().max().item() < 1e-6
    assert (rec.cam2world - traj_o).abs().max().item() < 1e-5
    assert rec.points.shape == cloud_o.shape
    assert (rec.points.cpu() - cloud_o).abs().max().item() < 1e-3 * cloud_o.abs().max().item()


def test_argument_errors():
    from coivo_amd import inference as I
    depth, K, M, _ = _scene(2, 8, 8, 80)
    d = to_dev({"depth": depth, "K": K, "M": M})
    with pytest.raises(ValueError):
        I.backproject(depth, K, M)                          # CPU tensors: no fallback
    with pytest.raises(ValueError):
        I.backproject(d["depth"], d["K"][:1], d["M"])
    with pytest.raises(ValueError):
        I.stitch_point_cloud(d["depth"], d["K"], d["M"], stride=0)
