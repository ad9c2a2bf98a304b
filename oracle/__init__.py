"""Test-only CPU oracle. See vae_oracle.py."""
