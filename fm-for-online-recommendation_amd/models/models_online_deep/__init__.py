"""models/models_online_deep: part of the drop-in mirror of the reference's import paths."""
