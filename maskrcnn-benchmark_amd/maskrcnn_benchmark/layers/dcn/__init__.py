"""Deformable convolution / pooling operators (reference layers/dcn/)."""
