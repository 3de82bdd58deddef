"""MI355X-native hot path of lucidrains/e2-tts-pytorch."""
