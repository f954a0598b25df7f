"""MI355X-native drop-in for BloomScene's depth-diff Gaussian rasterizer (hot path only)."""
