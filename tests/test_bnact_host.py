"""CPU: host logic of the (f4) BatchNorm passes - which maps count as channels_last rows (ops._rows_cl) and when the encoder's
bn_act takes the fused path (never for a CPU tensor: the torch sequence runs there, so the encoder stays importable and testable
without a GPU).  The kernels themselves: tests/test_gpu_bnact.py."""
import torch

from hoisdf_amd import ops
from hoisdf_amd.nets import encoder as E


def test_rows_view_of_channels_last_maps_and_channel_slices():
    x = torch.randn(2, 64, 6, 5).contiguous(memory_format=torch.channels_last)
    t, ld = ops._rows_cl(x)
    assert t.data_ptr() == x.data_ptr() and ld == 64
    wide = torch.randn(2, 160, 6, 5).contiguous(memory_format=torch.channels_last)
    sl = wide[:, 32:96]                                   # a channel slice keeps the parent's row stride: no copy
    t, ld = ops._rows_cl(sl)
    assert t.data_ptr() == sl.data_ptr() and ld == 160
    nchw = torch.randn(2, 64, 6, 5)                       # NCHW-contiguous: copied into rows
    t, ld = ops._rows_cl(nchw)
    assert ld == 64 and t.stride(1) == 1 and torch.equal(t, nchw)
    one = torch.randn(3, 32, 1, 1).contiguous(memory_format=torch.channels_last)      # strides of size-1 dims say nothing
    t, ld = ops._rows_cl(one)
    assert ld == 32 and torch.equal(t, one)
    odd = wide[:, 2:66]                                   # 8-byte aligned start: not usable in place (16-byte row loads)
    t, ld = ops._rows_cl(odd)
    assert t.data_ptr() != odd.data_ptr() and ld == 64 and torch.equal(t, odd)
    col = torch.randn(2, 64, 7, 1).contiguous(memory_format=torch.channels_last)      # W == 1: the row stride is H's
    t, ld = ops._rows_cl(col)
    assert ld == 64 and t.data_ptr() == col.data_ptr()


def test_supported_shapes():
    assert not ops.bn_act_supported(torch.randn(2, 64, 4, 4))                 # CPU tensor
    assert not ops.bn_act_supported(torch.randn(2, 64, 4, 4).double())


def test_cpu_encoder_runs_the_torch_sequence_whatever_the_switch_says():
    torch.manual_seed(0)
    bn = torch.nn.BatchNorm2d(16).train()
    x, res = torch.randn(3, 16, 5, 5), torch.randn(3, 16, 5, 5)
    want_bn = torch.nn.BatchNorm2d(16).train()
    want = torch.relu(want_bn(x) + res)
    for on in (True, False):
        E.set_bn_fused(on)
        bn.load_state_dict(torch.nn.BatchNorm2d(16).state_dict())
        got = E.bn_act(bn, x, True, residual=res)
        assert torch.allclose(got, want, atol=1e-6)
    E.set_bn_fused(True)
    E.flush_bn_counters()
    assert int(bn.num_batches_tracked) == 1 and torch.allclose(bn.running_mean, want_bn.running_mean)
