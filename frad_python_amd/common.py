"""Stream constants of the FrAD container (mirror of src/libfrad/common.py:1-10 of the reference)."""
SIGNATURE = b"fRad"
FRM_SIGN = b"\xff\xd0\xd2\x98"


def _crc16_table():
    tab = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0xA001 if c & 1 else c >> 1
        tab.append(c)
    return tab


_CRC16 = _crc16_table()


def crc16_ansi(data: bytes) -> int:
    crc = 0
    for b in data:
        crc = (crc >> 8) ^ _CRC16[(crc ^ b) & 0xFF]
    return crc
