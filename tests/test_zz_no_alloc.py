"""The allocation-free entry point mi355_guidance_rescale_static against mi355_guidance_rescale
(which takes a caller scratch buffer): same bits, on two calls in a row, the second on a
smaller input, so that whatever the first call left in the library's scratch array is shown not
to matter."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _pair(batch, tokens, channels, seed, device):
    g = torch.Generator().manual_seed(seed)
    cond = torch.randn(batch, tokens, channels, generator=g).bfloat16().to(device)
    noise = (torch.randn(batch, tokens, channels, generator=g) * 1.7 + 0.1).bfloat16().to(device)
    return noise, cond


def test_static_scratch_entry_point_equals_the_caller_scratch_one(device):
    from sglang_diffusion_amd import _C

    for batch, tokens, channels, seed in ((6, 4096, 64, 1), (2, 384, 64, 2)):   # second: smaller
        noise, cond = _pair(batch, tokens, channels, seed, device)
        want = noise.clone()
        _C.guidance_rescale(want, cond, 0.7)
        got = noise.clone()
        torch.cuda.synchronize()
        before = torch.cuda.memory_allocated(device)
        _C.guidance_rescale_static(got, cond, 0.7)
        torch.cuda.synchronize()
        assert torch.cuda.memory_allocated(device) == before
        assert not torch.equal(got, noise), "the rescale must change the tensor"
        assert torch.equal(got, want), (batch, tokens)
    # the limit is enforced with a clear message, and nothing is launched
    noise, cond = _pair(65, 8, 8, 3, device)
    keep = noise.clone()
    with pytest.raises(RuntimeError, match="exceeds the built-in scratch"):
        _C.guidance_rescale_static(noise, cond, 0.7)
    torch.cuda.synchronize()
    assert torch.equal(noise, keep)
    # exactly at the limit
    noise, cond = _pair(64, 16, 8, 4, device)
    want = noise.clone()
    _C.guidance_rescale(want, cond, 0.3)
    _C.guidance_rescale_static(noise, cond, 0.3)
    assert torch.equal(noise, want)
