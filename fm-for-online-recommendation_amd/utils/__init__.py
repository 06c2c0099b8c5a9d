"""utils: part of the drop-in mirror of the reference's import paths."""
