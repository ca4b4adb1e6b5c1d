"""feed_forward_vqgan_clip_amd — MI355X-native training hot path of feed-forward VQGAN-CLIP.

Host code is Python on PyTorch-ROCm (allocator, streams, autograd tape, torch.distributed);
all arithmetic of the hot path runs in hand-written HIP kernels for gfx950 behind the C ABI
declared in include/ffvc.h (feed_forward_vqgan_clip_amd/lib/libffvc_hip.so).
"""
__version__ = "0.1.0"
