"""MI355X-native GP fit path (drop-in for the hot path of Spatial_GP_repo/utils.py)."""
__all__ = ["synthetic", "engine", "build"]
