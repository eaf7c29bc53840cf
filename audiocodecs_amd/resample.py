"""Sample-rate conversion at the Codec boundary (codec.py:59-63,95-99 call
torchaudio.functional.resample).  Equal rates return the input unchanged -- exactly what
torchaudio does and all this round's configurations need (BASELINE.json feeds 24 kHz batches to a
24 kHz codec).  The polyphase windowed-sinc kernel for unequal rates is SURVEY.md §8(f1): next.
"""


def resample(sig, orig_freq, new_freq):
    if int(orig_freq) == int(new_freq):
        return sig
    raise NotImplementedError(
        f"resampling {orig_freq} -> {new_freq} Hz is not built yet (SURVEY.md §8 f1); "
        "construct the codec with sample_rate == orig_sample_rate"
    )
