// jackclient.h — JACK client base with the reference's surface
// (reference src/jackclient.h:10-63): name/handle/samplerate/ports, start/stop,
// isRunning, protected addInput/addOutput/activate and the four virtual hooks.
#pragma once
#include <jack/jack.h>
#include <jack/midiport.h>

#include <cassert>
#include <map>
#include <string>

#include "log.h"

typedef jack_port_t* JackPort;

class JackClient {
public:
    const std::string name;
    jack_client_t* handle = nullptr;
    size_t samplerate = 0;
    std::map<std::string, JackPort> ports;

    explicit JackClient(const std::string& name);
    virtual ~JackClient();

    void start();
    void stop();
    inline bool isRunning() const { return _isRunning; }

protected:
    JackPort addInput(const std::string& portName, const std::string& type = JACK_DEFAULT_AUDIO_TYPE, size_t bufferSize = 0);
    JackPort addOutput(const std::string& portName, const std::string& type = JACK_DEFAULT_AUDIO_TYPE, size_t bufferSize = 0);
    void activate();

    virtual void onStart() {}
    virtual void onStop() {}
    virtual void onProcess(size_t nframes) = 0;
    virtual void onShutdown() {}

private:
    bool _isRunning = false;
    JackPort addPort(const std::string& portName, const std::string& type, unsigned long flags, size_t bufferSize);
    static int processCallback(jack_nframes_t nframes, void* arg);
    static void shutdownCallback(void* arg);
};
