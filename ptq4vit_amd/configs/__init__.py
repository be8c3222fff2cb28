"""Quantisation configs with the reference's attribute names and ``get_module`` factory
(reference configs/PTQ4ViT.py, configs/BasePTQ.py)."""
