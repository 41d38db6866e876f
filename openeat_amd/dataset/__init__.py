"""Data side of the hot path: manifest -> batches -> rank split -> padded device batch.
Mirrors /root/reference/openeat/dataset/{dataset,audio_processor}.py; file storage formats beyond plain PCM wav and
uncompressed Kaldi float matrices, sox and text tokenisation are out of scope."""
