"""MI355X-native HierSpeech++ waveform generation (drop-in for the hot path of
liuhuang31/Megatts2_HierSpeechpp).  See DESIGN.md."""
__version__ = "0.1.0"
