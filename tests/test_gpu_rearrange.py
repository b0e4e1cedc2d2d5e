"""Row rearrangement on the device against a numpy statement of the reference's layouts
(merge.rs:224-358, permute_qk.rs:46-69): bytes must be identical."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def np_permute_qk(buf, row_bytes, rows, nh):
    dh = rows // nh
    x = buf.reshape(nh, 2, dh // 2, row_bytes)          # src row = h*dh + a*dh/2 + b
    return np.ascontiguousarray(x.transpose(0, 2, 1, 3)).reshape(-1)   # dst row = h*dh + 2b + a


@pytest.mark.parametrize("ty,row_elems", [(12, 4096), (2, 4096), (14, 2048), (0, 100), (1, 33), (8, 64), (24, 7)])
def test_permute_qk(ty, row_elems):
    import torch
    from gguf_amd import rearrange as R
    rb = R.row_bytes(ty, row_elems)
    for nh, dh in ((32, 128), (4, 64), (1, 2), (8, 6), (3, 10), (7, 14), (5, 96), (2, 250)):
        rows = nh * dh
        src = np.random.default_rng(rows + ty).integers(0, 256, rows * rb, dtype=np.uint8)
        got = R.permute_qk(ty, row_elems, rows, nh, torch.from_numpy(src).cuda())
        torch.cuda.synchronize()
        assert np.array_equal(got.cpu().numpy(), np_permute_qk(src, rb, rows, nh))


def test_merge_then_split_qkv_and_attn_qkv_permute():
    """merge_qkv -> (split, permute q and k, merge) as permute_qk.rs:30-35 does for attn_qkv."""
    import torch
    from gguf_amd import rearrange as R
    ty, c, nh, nkvh, dh = 12, 2048, 32, 4, 64
    rb = R.row_bytes(ty, c)
    rng = np.random.default_rng(1)
    q = rng.integers(0, 256, nh * dh * rb, dtype=np.uint8)
    k = rng.integers(0, 256, nkvh * dh * rb, dtype=np.uint8)
    v = rng.integers(0, 256, nkvh * dh * rb, dtype=np.uint8)
    dq, dk, dv = (torch.from_numpy(a).cuda() for a in (q, k, v))
    qkv = R.merge_qkv(ty, c, dq, dk, dv, nh * dh, nkvh * dh)
    torch.cuda.synchronize()
    assert np.array_equal(qkv.cpu().numpy(), np.concatenate([q, k, v]))
    rows = (nh + 2 * nkvh) * dh
    q2, k2, v2 = R.split_qkv(ty, c, qkv, rows, nh, nkvh)
    q3 = R.permute_qk(ty, c, nh * dh, nh, q2)
    k3 = R.permute_qk(ty, c, nkvh * dh, nkvh, k2)
    out = R.merge_qkv(ty, c, q3, k3, v2, nh * dh, nkvh * dh)
    torch.cuda.synchronize()
    want = np.concatenate([np_permute_qk(q, rb, nh * dh, nh), np_permute_qk(k, rb, nkvh * dh, nkvh), v])
    assert np.array_equal(out.cpu().numpy(), want)


def test_argument_checks():
    import torch
    import gguf_amd as g
    from gguf_amd import rearrange as R
    x = torch.zeros(144 * 64, dtype=torch.uint8, device="cuda")
    with pytest.raises(g.Indivisible):
        R.permute_qk(12, 100, 64, 4, x)        # row not a whole number of blocks
    with pytest.raises(g.Indivisible):
        R.permute_qk(12, 256, 63, 4, x)        # rows not a multiple of 2 * heads
    with pytest.raises(g.Unsupported):
        R.permute_qk(31, 256, 64, 4, x)        # Q4_0_4_4: no size in the format
