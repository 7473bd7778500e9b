"""Test-infrastructure stub for `timm` (reference swinir.py:15 imports three init-time helpers)."""
