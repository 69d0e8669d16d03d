#include "midi.h"

namespace {
// data bytes that follow a channel-voice status byte
int data_bytes(uint8_t status) {
    switch (status & 0xF0) {
        case 0xC0:
        case 0xD0: return 1;
        case 0x80:
        case 0x90:
        case 0xA0:
        case 0xB0:
        case 0xE0: return 2;
        default: return 0;
    }
}
}  // namespace

void RawMidi::Device::feed(const uint8_t* data, size_t len) {
    for (size_t i = 0; i < len; i++) {
        const uint8_t b = data[i];
        if (b & 0x80) {
            if (b >= 0xF8) continue;  // real-time bytes do not disturb running status
            runningStatus = (b < 0xF0) ? b : 0;
            pending.clear();
            continue;
        }
        if (!runningStatus) continue;
        pending.push_back(b);
        if ((int)pending.size() == data_bytes(runningStatus)) {
            uint8_t msg[3] = {runningStatus, pending[0], pending.size() > 1 ? pending[1] : (uint8_t)0};
            if (handler) handler->onMidiMessage(this, msg, 1 + pending.size());
            pending.clear();
        }
    }
}
