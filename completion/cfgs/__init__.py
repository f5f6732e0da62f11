"""YAML configurations of the completion entry points."""
