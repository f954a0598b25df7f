"""Forms kept for measurement on hardware the product path has not seen yet; nothing here is imported by default."""
