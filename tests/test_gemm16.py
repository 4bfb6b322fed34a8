"""GPU: ww_gemm16_nt -- the 16-bit-operand GEMM core (operands resident in HBM as bf16 / fp16, fp32 accumulation) against
float64 torch on the SAME 16-bit operand values: the products are exact in fp32, so the only error is the accumulation
order (bound 2e-6 of the row scale at K = 1024); 16-bit output = that result rounded once."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def nat():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from wakeword_trainer_home_amd import _native
    _native.load()
    return _native


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 128), (1, 1, 64), (130, 70, 192), (1000, 513, 1024), (4096, 1024, 576),
                                   (127, 129, 64)])
def test_gemm16_nt_matches_float64(nat, dtype, M, N, K):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    a = (torch.randn(M, K, generator=g) * 0.5).to(dtype)
    b = (torch.randn(N, K, generator=g) * 0.5).to(dtype)
    ref = a.double() @ b.double().T
    scale = ref.abs().max().item() + 1e-30
    out = nat.gemm16_nt(a.to(DEV), b.to(DEV))
    assert out.dtype == torch.float32 and out.shape == (M, N)
    assert (out.cpu().double() - ref).abs().max().item() <= 3e-6 * scale
    out16 = nat.gemm16_nt(a.to(DEV), b.to(DEV), out_dtype=dtype)
    assert out16.dtype == dtype
    assert torch.equal(out16.cpu(), out.cpu().to(dtype))          # the fp32 result rounded once (RNE)


def test_gemm16_nt_rejects_bad_input(nat):
    a = torch.zeros(64, 64, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(ValueError):
        nat.gemm16_nt(a, torch.zeros(64, 64, dtype=torch.float16, device=DEV))     # mixed operand types
    with pytest.raises(ValueError):
        nat.gemm16_nt(a.float(), a.float())                                          # fp32 operands belong to ww_linear_mfma_*
    with pytest.raises(nat.NativeError):
        nat.gemm16_nt(torch.zeros(64, 96, dtype=torch.bfloat16, device=DEV), torch.zeros(8, 96, dtype=torch.bfloat16, device=DEV))  # K % 64
    with pytest.raises(ValueError):
        nat.gemm16_nt(a, a[:, :32])
