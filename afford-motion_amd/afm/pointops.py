"""placeholder - filled in with the point-cloud kernels (FPS, kNN, fused set abstraction, vector attention)."""
