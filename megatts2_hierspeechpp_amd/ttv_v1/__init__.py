"""HIP mirror of the reference's ``ttv_v1`` package (text -> wav2vec front-end and the Mega-TTS2
prosody language model): same class names, constructor arguments and state-dict keys."""
