"""Counterparts of the reference's utils package: models, net_wrap, quant_calib, integer."""
