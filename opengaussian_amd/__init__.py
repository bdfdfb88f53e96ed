"""MI355X-native rasterizer + k-means hot path for OpenGaussian."""
