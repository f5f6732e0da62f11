"""Completion harness (train.py / test.py entry points, models, cfgs)."""
