"""Fiat-Shamir transcript with the byte-level behaviour of the reference's transcript.py
(SURVEY.md 8f N4), so provers/verifiers built on this package derive challenges the same way:

  state_0          = SHA256(label)                                        transcript.py:23
  append(l, data)  : state = SHA256(state || l || ser(data))              transcript.py:95-100
  challenge(l)     : h = SHA256(state || l); c = F(int(h, big-endian));
                     state = SHA256(state || l || h); return c            transcript.py:47-56
  ser(str) = utf-8;  ser(int) = 8-byte big-endian signed (struct ">q");
  ser(bytes) = itself;  ser(list) = concatenation of ser(items);
  anything else = str(x).encode()   (field elements, point tuples)        transcript.py:66-85

Host-side, bytes-sized work; not on the GPU path."""
import hashlib
import struct


def _sha(*parts):
    h = hashlib.sha256()
    for p in parts:
        h.update(p)
    return h.digest()


class Transcript:
    def __init__(self, label, F):
        self.label = label
        self.F = F
        self.state = _sha(label.encode())

    @staticmethod
    def _serialize(data):
        if isinstance(data, str):
            return data.encode()
        if isinstance(data, int):
            return struct.pack(">q", data)
        if isinstance(data, bytes):
            return data
        if isinstance(data, list):
            return b"".join(Transcript._serialize(item) for item in data)
        return str(data).encode()

    def append_message(self, message_label, message_data):
        self.state = _sha(self.state, message_label.encode(), self._serialize(message_data))

    def get_challenge(self, label):
        digest = _sha(self.state, label.encode())
        challenge = self.F(int.from_bytes(digest, byteorder="big"))
        self.state = _sha(self.state, label.encode(), digest)
        return challenge
