"""Drop-in counterparts of the reference's quant_layers package (linear / matmul / conv).

Same class names, constructor keywords, modes and state attributes as hahnyuan/PTQ4ViT's
``quant_layers``; ``calibration_step2`` runs on the MI355X through the C ABI
in include/ptq4vit_hip.h instead of a sequence of torch ops.
"""
