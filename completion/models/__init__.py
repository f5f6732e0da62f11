"""Completion networks: pcn, ecg, vrcnet (train.py imports `models.<model_name>`)."""
