"""MI355X-native hot path of the stereo-depth + point-cloud fusion pipeline."""
