"""Chamfer distance operator (dist_chamfer_3D.chamfer_3DDist)."""
