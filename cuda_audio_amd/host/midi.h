// midi.h — RawMidi::MessageHandler / Device with the reference's surface
// (reference src/midi.h:16-46).  ALSA is not available here, so a Device has
// no reader thread: bytes are pushed with feed(), which performs the same
// running-status reassembly as the reference's reader (midi.cu:22-59) and
// dispatches complete messages to the handler.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

class RawMidi {
public:
    class Device;

    class MessageHandler {
    public:
        virtual ~MessageHandler() {}
        virtual void onMidiMessage(const Device* sender, const uint8_t* buffer, size_t len) = 0;
    };

    class Device {
    public:
        explicit Device(const std::string& deviceId) : id(deviceId) {}
        virtual ~Device() {}
        void start() { isOpen = isRunning = true; }
        void stop() { isOpen = isRunning = false; }
        // push raw MIDI bytes (status bytes optional after the first: running status)
        void feed(const uint8_t* data, size_t len);

        MessageHandler* handler = nullptr;
        std::string id;
        bool isOpen = false, isRunning = false;
        uint8_t runningStatus = 0;

    private:
        std::vector<uint8_t> pending;
    };
};
