"""Pins the oracle network (oracle/net_ref.py):
  * against the committed golden vectors, which were produced by the REFERENCE's
    own torch modules (oracle/gen_golden.py) -- runs everywhere;
  * bit-for-bit against the reference modules themselves when /root/reference
    is present (build container only)."""
import numpy as np
import pytest
import torch

from conftest import checkpoint, ckpt_checksum, load_golden
from oracle import gen_golden
from oracle.net_ref import OracleNet
from oracle.ref_import import reference_available

SMALL = sorted(gen_golden.SMALL_CASES)


@pytest.mark.parametrize("name", SMALL)
def test_oracle_matches_reference_golden(name):
    g = load_golden(name)
    ck = checkpoint(int(g["wseed"]))
    # the seeded generators must reproduce the exact weights/inputs the fixture was made from
    assert ckpt_checksum(ck) == pytest.approx(float(g["ckpt_sum"]), rel=0, abs=1e-9)
    x = gen_golden.make_input(int(g["iseed"]), tuple(int(v) for v in g["shape"]))
    assert float(x.double().sum()) == pytest.approx(float(g["input_sum"]), rel=0, abs=1e-9)
    blks, mask, lines = OracleNet(ck)(x)
    # same torch build + same op sequence => identical bits; allow 2 ulp-ish slack for
    # a different CPU's oneDNN kernel selection on the GPU box's host
    np.testing.assert_allclose(mask.numpy(), g["mask"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(lines.numpy(), g["lines"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(blks.numpy(), g["blks"], rtol=2e-5, atol=2e-4)


@pytest.mark.skipif(not reference_available(), reason="reference tree only exists in the build container")
def test_oracle_bit_exact_vs_reference_modules():
    from oracle.ref_import import ReferenceNet
    ck = checkpoint(0)
    x = torch.rand(2, 3, 128, 192, generator=torch.Generator().manual_seed(7))
    rb, rm, rl = ReferenceNet(ck)(x)
    ob, om, ol = OracleNet(ck)(x)
    assert torch.equal(rb, ob) and torch.equal(rm, om) and torch.equal(rl, ol)


@pytest.mark.skipif(not reference_available(), reason="reference tree only exists in the build container")
def test_synthetic_checkpoint_is_in_reference_format():
    """strict load_state_dict into the reference's Model / UnetHead / DBHead succeeds."""
    from oracle.ref_import import ReferenceNet
    ref = ReferenceNet(checkpoint(3))
    n = sum(p.numel() for m in (ref.blk_det, ref.text_seg, ref.text_det) for p in m.parameters())
    assert 23.0e6 < n < 23.8e6     # SURVEY: 23.4 M params


@pytest.mark.skipif(not reference_available(), reason="reference tree only exists in the build container")
def test_oracle_step_function_equals_reference_dbhead_step_eval():
    """`DBHead.forward(step_eval=True)` (reference basemodel.py:121-122,159-160) vs the restated step function
    applied to the oracle's two DB planes."""
    from oracle.ref_import import ReferenceNet
    ck = checkpoint(0)
    x = torch.rand(1, 3, 128, 128, generator=torch.Generator().manual_seed(11))
    ref = ReferenceNet(ck)
    with torch.no_grad():
        blks, feats = ref.blk_det(x, detect=True)
        mask, feats2 = ref.text_seg(*feats, forward_mode=ref._mode)
        theirs = ref.text_det(*feats2, step_eval=True)
    _, _, lines = OracleNet(ck)(x)
    ours = OracleNet.step_function(lines, k=ref.text_det.k)
    assert theirs.shape == ours.shape == (1, 1, 128, 128)
    assert torch.equal(theirs, ours)
