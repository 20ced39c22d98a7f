"""MI355X-native fine-tuning hot path for N-Best-ASR-Transformer (gfx950 / CDNA4).

Host side is Python on PyTorch-ROCm (device memory, streams, torch.distributed = RCCL); all
arithmetic on the path is hand-written HIP behind the C-ABI library ``csrc/libnbest_hip.so``
(declared in ``include/nbest_hip.h``).  Import as ``import nbest_amd``.

Submodules
  config     encoder / label-space configuration
  synth      deterministic synthetic weights and n-best batches (bench + parity tests)
  hipabi     ctypes binding of the C-ABI (fails loudly when the library is missing)
  arena      flat fp32 master / grad / bf16 compute parameter arenas with HF-named views
  model      TOD_ASR_Transformer_STC-compatible module whose encoder is the HIP path
  trainer    train_epoch / eval_epoch / data-parallel step
  inputs     prepare_inputs_for_roberta-compatible host input builder
  fscore     update_f1 / compute_f1
"""
__version__ = "0.1.0"
