#include "jackclient.h"

JackClient::JackClient(const std::string& n) : name(n) {}
JackClient::~JackClient() {}

int JackClient::processCallback(jack_nframes_t nframes, void* arg) {
    // reference src/jackclient.cu:4-11: always reports success to the server
    assert(arg);
    static_cast<JackClient*>(arg)->onProcess(nframes);
    return 0;
}

void JackClient::shutdownCallback(void* arg) {
    auto* self = static_cast<JackClient*>(arg);
    Log::warn(self->name, "JACK server is shutting down");
    self->onShutdown();
}

void JackClient::start() {
    // reference src/jackclient.cu:24-44: open (never start a server), callbacks, rate, then onStart()
    jack_status_t status = 0;
    Log::info(name, "opening JACK client");
    handle = jack_client_open(name.c_str(), JackNoStartServer, &status, nullptr);
    assert(handle && "jack_client_open failed");
    assert(!(status & JackNameNotUnique));
    jack_set_process_callback(handle, &JackClient::processCallback, this);
    jack_on_shutdown(handle, &JackClient::shutdownCallback, this);
    samplerate = jack_get_sample_rate(handle);
    Log::info(name, "sample rate %zu", samplerate);
    _isRunning = true;
    onStart();
}

void JackClient::stop() {
    assert(handle && _isRunning);
    onStop();
    jack_client_close(handle);
    handle = nullptr;
    _isRunning = false;
}

JackPort JackClient::addPort(const std::string& portName, const std::string& type, unsigned long flags, size_t bufferSize) {
    JackPort p = jack_port_register(handle, portName.c_str(), type.c_str(), flags, bufferSize);
    assert(p && "jack_port_register failed");
    ports[portName] = p;
    return p;
}

JackPort JackClient::addInput(const std::string& portName, const std::string& type, size_t bufferSize) {
    Log::info(name, "input port %s", portName.c_str());
    return addPort(portName, type, JackPortIsInput, bufferSize);
}

JackPort JackClient::addOutput(const std::string& portName, const std::string& type, size_t bufferSize) {
    Log::info(name, "output port %s", portName.c_str());
    return addPort(portName, type, JackPortIsOutput, bufferSize);
}

void JackClient::activate() {
    jack_activate(handle);
    Log::info(name, "activated");
}
