"""Pure-Python builders for framed streams used by the parity tests: an INDEPENDENT restatement of the container
grammar (snappy framing_format.txt) so that hand-made streams do not come from the code under test."""
import struct

SNAPPY_IDENT = b"\xff\x06\x00\x00sNaPpY"


def crc32c(data):
    """bitwise CRC-32C (Castagnoli), no tables"""
    c = 0xFFFFFFFF
    for b in bytes(data):
        c ^= b
        for _ in range(8):
            c = (c >> 1) ^ (0x82F63B78 if c & 1 else 0)
    return c ^ 0xFFFFFFFF


def crc32c_masked(data):
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def snappy_chunk(ty, body):
    return bytes([ty]) + struct.pack("<I", len(body))[:3] + body


def snappy_stored(piece, crc=None):
    return snappy_chunk(0x01, struct.pack("<I", crc32c_masked(piece) if crc is None else crc) + bytes(piece))


def snappy_compressed(piece, raw_block, crc=None):
    return snappy_chunk(0x00, struct.pack("<I", crc32c_masked(piece) if crc is None else crc) + bytes(raw_block))
