"""Empty stand-in for open3d (imported, never called on the trainer path)."""
