"""Prompt denoiser of the TTS harness (reference: denoiser/, MP-SENet)."""
