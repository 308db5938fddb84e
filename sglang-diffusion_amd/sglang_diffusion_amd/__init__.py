"""MI355X-native DiT hot path of SGLang-Diffusion (Flux.1 denoise step).

Host-side mirror of the reference's plugin interfaces (attention backend,
quantization method, linear layers, Flux DiT, flow-match Euler scheduler,
denoise loop) over the C-ABI kernel library `libmi355dit.so`.
"""

__version__ = "0.1.0"
